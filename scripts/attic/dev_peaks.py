"""Measured stream / MFMA ceilings (dev helper around bench.measured_peaks)."""
import json, sys
import torch
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.measured_peaks(torch.device("cuda:0"))))

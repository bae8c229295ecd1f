"""Per-iteration times of a Lloyd run at group_paras.py's default shape (HIP-event time of the assignment, wall time of the
whole iteration) -- e.g. PROQA_KMEANS_SORTED=0/1, PROQA_KMEANS_TWO_PASS=0/1."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from proqa_amd.group_paras import KMeans
dev = torch.device("cuda:0")
n, k = 10_000_000, 10_000
x = bench.gen_rows(0, n, dev)
km = KMeans(128, k, niter=10, max_points_per_centroid=n // k + 1, verbose=False)
km.train(x)
torch.cuda.synchronize()
print("assign ms :", " ".join(f"{v:6.1f}" for v in km.assign_ms))
print("iteration :", " ".join(f"{v * 1e3:6.1f}" for v in km.iter_seconds), " objective", f"{km.obj[-1]:.9e}")

"""Single-query latency (qa/online_sampler.py:104-121 shape: one question at a time) over 18M rows."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((min(2_000_000, n - r0), 128), generator=g, device=dev).half()
index = IndexFlatIP(128); index.adopt_device(xb)
for nq in (1, 8, 32, 128, 256, 512, 1024, 2032, 4096):
    xq = torch.randn((nq, 128), generator=g, device=dev).half()
    for k in (5, 80, 1000, 5000):
        if nq > 32 and k > 80:
            continue
        for _ in range(2):
            index.search_device(xq, k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            D, I = index.search_device(xq, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"nq={nq:5d} k={k:5d}: {dt*1e3:8.3f} ms  {nq/dt:10.0f} q/s  scan {n*256/dt/1e9:7.0f} GB/s-equiv")

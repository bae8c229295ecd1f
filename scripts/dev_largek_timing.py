"""Developer timing of the paged large-k search at the trec_process.py shape (8.8M passages, k=10000)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n, nq, k = 8_800_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 10000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
ix.search_device(xq, k)
torch.cuda.synchronize()
t = time.time()
D, I = ix.search_device(xq, k)
torch.cuda.synchronize()
dt = time.time() - t
import os
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t = time.time(); D, I = ix.search_device(xq, k); torch.cuda.synchronize(); ts.append(time.time() - t)
print("flags", os.environ.get("PROQA_FILTER_FLAGS"), "ms", [round(x * 1e3, 2) for x in ts], "digest", int(I.sum().item()), float(D.double().sum().item()))
print(f"n={n} nq={nq} k={k}: {dt*1e3:.1f} ms ({nq/dt:.0f} q/s), stats {ix.last_stats()}")

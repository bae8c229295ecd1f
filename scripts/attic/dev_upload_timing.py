"""Host -> HBM upload rate of IndexFlatIP.add (the PCIe-inclusive side of the boundary)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(0)
xb = rng.standard_normal((n, 128), dtype=np.float32).astype(np.float16)
xq = rng.standard_normal((2032, 128), dtype=np.float32).astype(np.float16)
for dtype in (np.float16, np.float32):
    x = xb.astype(dtype)
    for rep in range(3):
        index = IndexFlatIP(128, capacity=n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index.add(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"add {dtype.__name__} {n} rows: {dt*1e3:.1f} ms = {x.nbytes/dt/1e9:.2f} GB/s host bytes")
    t0 = time.perf_counter()
    D, I = index.search(xq, 80)
    dt = time.perf_counter() - t0
    print(f"search (host in/out) 2032 q: {dt*1e3:.2f} ms")
    t0 = time.perf_counter()
    D, I = index.search(xq, 80)
    dt = time.perf_counter() - t0
    print(f"search (host in/out) 2032 q: {dt*1e3:.2f} ms")

"""Developer timing of the HIP search at the BASELINE shape (not part of the test suite)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2032
k = 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
ix.set_profiling(True)
for cfg in [(2048, 256, 4), (2048, 256, 2), (2048, 256, 8), (1024, 256, 4)]:
    ix.configure(*cfg)
    for _ in range(2):
        ix.search_device(xq, k)
    torch.cuda.synchronize()
    t = time.time()
    reps = 5
    for _ in range(reps):
        D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    st = ix.last_stats()
    flops = 2.0 * nq * n * 128
    print(f"cfg={cfg} wall={dt*1e3:.3f} ms  q/s={nq/dt:.0f}  stats={st}  "
          f"filter TF/s={flops/st['filter_ms']/1e9:.1f}  HBM-equivalent GB/s={n*256/st['filter_ms']/1e6:.0f}")

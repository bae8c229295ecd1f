"""Developer timing of the HIP search at the BASELINE shape (not part of the test suite)."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2032
cfgs = [(256, 4)] if len(sys.argv) <= 3 else [tuple(int(x) for x in c.split(",")) for c in sys.argv[3:]]
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
for cfg in cfgs:
    ix.configure(*cfg)
    for _ in range(3):
        ix.search_device(xq, k)
    torch.cuda.synchronize()
    reps = 10
    best = 1e9
    t = time.time()
    for _ in range(reps):
        D, I = ix.search_device(xq, k)
        best = min(best, ix.last_stats()["filter_ms"])
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    st = ix.last_stats()
    flops = 2.0 * nq * n * 128
    print(f"variant={os.environ.get('PROQA_FILTER_VARIANT','default')} cfg={cfg} wall={dt*1e3:.3f} ms q/s={nq/dt:.0f} "
          f"rounds={st['rounds']} cand={st['candidates']} filter_ms={st['filter_ms']:.3f} (best {best:.3f}) total_ms={st['total_ms']:.3f} "
          f"filter TF/s={flops/st['filter_ms']/1e9:.1f}")

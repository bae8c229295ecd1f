"""Wall-clock time of whole searches (no per-launch profiling): rows, queries, [k] (dev helper)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nqs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2032]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
import os
ix = IndexFlatIP(128)
ix.adopt_device(xb)
if os.environ.get("BOOT"):
    ix.configure_bootstrap(int(os.environ["BOOT"]))
if os.environ.get("GROWTH"):
    ix.configure(0, int(os.environ["GROWTH"]))
for nq in nqs:
    xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
    for _ in range(3):
        ix.search_device(xq, k)
    torch.cuda.synchronize()
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    st = ix.last_stats()
    print(f"rows={n} nq={nq} k={k}: wall {dt*1e3:.3f} ms  {nq/dt:.0f} q/s  rounds={st['rounds']} fallback={st['fallback_rounds']} "
          f"cand/q={st['candidates']/nq:.0f}")

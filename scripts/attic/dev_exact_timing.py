"""Exact-float32 mode at the BASELINE shape: float32 corpus / queries that fp16 cannot hold (dev timing)."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2032
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
index = IndexFlatIP(128, capacity=n)
t0 = time.perf_counter()
for r0 in range(0, n, 2_000_000):
    index.add(torch.randn((min(2_000_000, n - r0), 128), generator=g, device=dev))
torch.cuda.synchronize()
print(f"add {n} float32 rows (device): {time.perf_counter()-t0:.2f} s, exact_f32={index.exact_f32}")
xq = torch.randn((nq, 128), generator=g, device=dev)
for k in (80,):
    for _ in range(2):
        index.search_device(xq, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        D, I = index.search_device(xq, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    st = index.last_stats()
    print(f"exact nq={nq} k={k}: {dt*1e3:.2f} ms  {nq/dt:.0f} q/s  rounds={st['rounds']} fallback={st['fallback_rounds']} cand/q={st['candidates']/nq:.0f}")
index.set_profiling(True); index.search_device(xq, 80); print("filter_ms", index.last_stats()["filter_ms"], "total_ms", index.last_stats()["total_ms"])

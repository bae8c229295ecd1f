"""dev: stream time of one search for the growth list in PROQA_GROWTH_LIST (read by the library at load)."""
import os, sys
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
n = int(float(sys.argv[1]))
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((2032, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128); ix.adopt_device(xb)
for _ in range(3): ix.search_device(xq, 80)
ts = []
for _ in range(10):
    ix.search_device(xq, 80); ts.append(ix.last_stats()["total_ms"])
st = ix.last_stats()
print(f"rows {n} list {os.environ.get('PROQA_GROWTH_LIST','-')}: stream {sorted(ts)[len(ts)//2]:.3f} ms rounds {st['rounds']} fallback {st['fallback_rounds']} cand/q {st['candidates']/2032:.0f}")

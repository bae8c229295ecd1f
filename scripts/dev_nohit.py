"""dev: filter time with the threshold test but without a single hit (PROQA_DEBUG_NOHIT=1, bootstrap off) next to the normal search."""
import os, sys
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
n = 18_000_000
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((2_000_000, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((2032, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128); ix.adopt_device(xb); ix.configure_bootstrap(0); ix.set_profiling(True)
for _ in range(3): ix.search_device(xq, 80)
f = []
for _ in range(8):
    ix.search_device(xq, 80); f.append(ix.last_stats()["filter_ms"])
st = ix.last_stats()
print(f"nohit={os.environ.get('PROQA_DEBUG_NOHIT','0')}: filter {sorted(f)[len(f)//2]:.3f} ms rounds {st['rounds']} cand/q {st['candidates']/2032:.0f} -> {2*2032*n*128/sorted(f)[len(f)//2]/1e9:.0f} TF/s")

import sys, torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128); ix.adopt_device(xb)
for i in range(3):
    ix.search_device(xq, 80)
    print(i, ix.last_stats(), flush=True)

import sys, time, torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
n, nq, k = 18_000_000, 2032, 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((min(2_000_000, n - r0), 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128); ix.adopt_device(xb)
for prof in (True, False, True, False):
    ix.set_profiling(prof)
    for _ in range(3): ix.search_device(xq, k)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): ix.search_device(xq, k)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print("profiling", prof, f"wall {dt*1e3:.3f} ms", ix.last_stats()["total_ms"])

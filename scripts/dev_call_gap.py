"""Where the time between two call-by-call searches goes: wall time per call against the GPU-side span of the search
(first to last kernel, from the library's own events), and the host's time inside / outside the call."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP

dev = torch.device("cuda:0")
n, nq, k = 18_000_000, 2032, 80
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((2_000_000, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
for rows in (18_000_000, 2_250_000):
    ix = IndexFlatIP(128); ix.adopt_device(xb[:rows])
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    for _ in range(5): ix.search_device(xq, k, out=out)
    spans, calls = [], []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        a = time.perf_counter(); ix.search_device(xq, k, out=out); b = time.perf_counter()
        calls.append(b - a); spans.append(ix.last_stats()["total_ms"])
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 40 * 1e3
    print(f"rows {rows}: wall per call {wall:.3f} ms; inside the call {sum(calls) / 40 * 1e3:.3f} ms; GPU span of a search "
          f"(first kernel start to last kernel end) {sum(spans) / 40:.3f} ms")
    ix.close()

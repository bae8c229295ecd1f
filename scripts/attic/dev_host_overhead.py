"""dev: where the wall time of one search call goes beyond its stream time (2.25M rows, 2032 queries)."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib
from proqa_amd.index import IndexFlatIP
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_250_000
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((2032, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128); ix.adopt_device(xb)
lib = _lib.load()
D = torch.empty((2032, 80), dtype=torch.float32, device=dev); I = torch.empty((2032, 80), dtype=torch.int64, device=dev)
st = _lib.current_stream_ptr()
for _ in range(5): ix.search_device(xq, 80)
torch.cuda.synchronize()
reps = 200
t0 = time.perf_counter()
for _ in range(reps): ix.search_device(xq, 80)
t_wrap = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps): ix.search_device(xq, 80, out=(D, I))
t_out = (time.perf_counter() - t0) / reps
t0 = time.perf_counter(); tot = 0.0
for _ in range(reps):
    lib.proqa_index_search_device(ix._h, xq.data_ptr(), 2032, 0, 80, 0, D.data_ptr(), I.data_ptr(), st)
    tot += ix.last_stats()["total_ms"]
t_raw = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    lib.proqa_index_search_device(ix._h, xq.data_ptr(), 2032, 0, 80, 0, D.data_ptr(), I.data_ptr(), st)
t_raw2 = (time.perf_counter() - t0) / reps
print(f"rows {n}: wrapper {t_wrap*1e3:.3f} ms | wrapper with out= {t_out*1e3:.3f} | raw C call + last_stats {t_raw*1e3:.3f} | raw C call {t_raw2*1e3:.3f} | stream (events) {tot/reps:.3f} ms")

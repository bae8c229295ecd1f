"""How much of a search is the host catching up after the synchronisation that ended the previous one?  The same 2032-query
search (k = 80) timed (a) call by call -- each call ends with its one host synchronisation -- and (b) with two index handles
over the same rows taking turns, begin / finish deferred so that the GPU always has the next search queued."""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib
from proqa_amd.index import IndexFlatIP, _torch_dtype_code

dev = torch.device("cuda:0")
n, nq, k = 18_000_000, 2032, 80
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((2_000_000, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
lib = _lib.load()
for rows in ([int(a) for a in sys.argv[1:]] or [18_000_000, 4_500_000, 2_250_000]):
    hs = [IndexFlatIP(128), IndexFlatIP(128)]
    for h in hs: h.adopt_device(xb[:rows])
    D = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(2)]
    I = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(2)]
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5): hs[0].search_device(xq, k)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): hs[0].search_device(xq, k)
    torch.cuda.synchronize(); sync_ms = (time.perf_counter() - t) / 40 * 1e3
    def begin(j):
        _lib.check(lib.proqa_index_search_begin_device(hs[j]._h, xq.data_ptr(), nq, _torch_dtype_code(xq), k, 0, D[j].data_ptr(),
                                                       I[j].data_ptr(), status[j:].data_ptr(), st))
    def finish(j):
        r = ctypes.c_int(); _lib.check(lib.proqa_index_search_finish(hs[j]._h, ctypes.byref(r)))
    begin(0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(40):
        begin((i + 1) & 1)      # the next search is queued ...
        finish(i & 1)           # ... before the host waits for this one
    finish(40 & 1)
    torch.cuda.synchronize(); pipe_ms = (time.perf_counter() - t) / 41 * 1e3
    print(f"rows {rows}: call by call {sync_ms:.3f} ms per search; next search queued before the wait {pipe_ms:.3f} ms "
          f"({(sync_ms - pipe_ms) * 1e3:.0f} us per search are the host catching up)")
    for h in hs: h.close()

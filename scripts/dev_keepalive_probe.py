"""Is the GPU slower after the short idle between two call-by-call searches (clock management), or is it only the idle time?
Call-by-call searches (begin / finish on one handle) with and without a one-wave spin kernel on a side stream that starts when
a search ends and bridges the host's turnaround."""
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
side = torch.cuda.Stream(device=dev)
for rows in (18_000_000, 2_250_000):
    ix = IndexFlatIP(128); ix.adopt_device(xb[:rows])
    D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream()
    def search(bridge_us):
        _lib.check(lib.proqa_index_search_begin_device(ix._h, xq.data_ptr(), nq, _torch_dtype_code(xq), k, 0, D.data_ptr(), I.data_ptr(),
                                                       status.data_ptr(), cur.cuda_stream))
        if bridge_us:
            ev = torch.cuda.Event(); ev.record(cur)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(bridge_us * 2100))     # ~2.1 GHz reference clock
        r = ctypes.c_int(); _lib.check(lib.proqa_index_search_finish(ix._h, ctypes.byref(r)))
    for bridge in (0, 150, 0, 150, 400):
        for _ in range(5): search(bridge)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): search(bridge)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 40 * 1e3
        print(f"rows {rows}: call by call, spin kernel of {bridge} us behind every search: {ms:.3f} ms per search")
    ix.close()

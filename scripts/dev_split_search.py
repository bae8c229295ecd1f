"""Prototype: the 2032-query search as two half-batches on two streams (two index handles over the same rows), so that the
merges of one half run beside the filter launches of the other.  Compares with the plain search, ids and scores included."""
import sys, time, ctypes
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib
from proqa_amd.index import IndexFlatIP, _torch_dtype_code

dev = torch.device("cuda:0")
n_all, nq, k = 18_000_000, 2032, 80
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.empty((n_all, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n_all, 2_000_000):
    m = min(2_000_000, n_all - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
lib = _lib.load()

def timed(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

ROWS = [int(a) for a in sys.argv[1:]] or [18_000_000, 9_000_000, 4_500_000, 2_250_000]
for rows in ROWS:
    part = xb[:rows]
    one = IndexFlatIP(128); one.adopt_device(part)
    D0, I0 = one.search_device(xq, k)
    base = timed(lambda: one.search_device(xq, k))
    halves = [IndexFlatIP(128), IndexFlatIP(128)]
    for h in halves: h.adopt_device(part)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    cut = 1024
    xqs = [xq[:cut].contiguous(), xq[cut:].contiguous()]
    D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    Ds = [D[:cut], D[cut:]]; Is = [I[:cut], I[cut:]]
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    for stagger in (False, True):
        def split():
            cur = torch.cuda.current_stream()
            ev0 = torch.cuda.Event(); ev0.record(cur)
            for j in range(2):
                streams[j].wait_event(ev0)
                with torch.cuda.stream(streams[j]):
                    # B starts when A's first large launch is under way: approximated by a short sleep kernel on B
                    if stagger and j == 1: torch.cuda._sleep(int(stagger_cycles))
                    _lib.check(lib.proqa_index_search_begin_device(halves[j]._h, xqs[j].data_ptr(), xqs[j].shape[0], _torch_dtype_code(xq), k, 0,
                                                                   Ds[j].data_ptr(), Is[j].data_ptr(), status[j:].data_ptr(),
                                                                   streams[j].cuda_stream))
            for j in range(2):
                r = ctypes.c_int()
                _lib.check(lib.proqa_index_search_finish(halves[j]._h, ctypes.byref(r)))
                cur.wait_stream(streams[j])
        stagger_cycles = 0.15e-3 * 2.0e9 * rows / 18_000_000 * 2   # ~ half of a half-round
        t = timed(split)
        same = bool((I == I0).all()) and bool((D == D0).all())
        print(f"rows {rows}: plain {base:.3f} ms, two halves on two streams (stagger={stagger}) {t:.3f} ms, identical={same}, "
              f"stats {halves[0].last_stats()['rounds']} rounds")
    one.close()
    for h in halves: h.close()

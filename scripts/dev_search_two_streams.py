"""Round-5 review, next #2(ii): a stream of query batches searched by TWO index handles over the same rows on TWO streams --
search i+1 is begun on the other stream before search i is finished, so that one search's bootstrap / small rounds / merges run
beside the other's large scans.  Against the call-by-call search (dev; MI355X).

    python scripts/dev_search_two_streams.py [rows ...]"""
import ctypes, hashlib, sys, time
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
dig = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]
for rows in ([int(float(a)) for a in sys.argv[1:]] or [18_000_000, 9_000_000, 4_500_000, 2_250_000]):
    hs = [IndexFlatIP(128), IndexFlatIP(128)]
    for h in hs:
        h.adopt_device(xb[:rows])
        h.prepare()
    D = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(2)]
    I = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(2)]
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(5):
        Dr, Ir = hs[0].search_device(xq, k)
        hs[1].search_device(xq, k)
    ref = dig(Ir) + dig(Dr)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): hs[0].search_device(xq, k)
    torch.cuda.synchronize(); sync_ms = (time.perf_counter() - t) / 40 * 1e3
    def begin(j):
        _lib.check(lib.proqa_index_search_begin_device(hs[j]._h, xq.data_ptr(), nq, _torch_dtype_code(xq), k, 0, D[j].data_ptr(),
                                                       I[j].data_ptr(), status[4 * j:].data_ptr(), streams[j].cuda_stream))
    def finish(j):
        r = ctypes.c_int(); _lib.check(lib.proqa_index_search_finish(hs[j]._h, ctypes.byref(r)))
    for reps in range(2):
        begin(0)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(40):
            begin((i + 1) & 1)      # the next search is on the other stream ...
            finish(i & 1)           # ... before the host waits for this one
        finish(40 & 1)
        torch.cuda.synchronize(); pipe_ms = (time.perf_counter() - t) / 41 * 1e3
    ok = (dig(I[0]) + dig(D[0]) == ref) and (dig(I[1]) + dig(D[1]) == ref)
    print(f"rows {rows}: call by call {sync_ms:.3f} ms per search; two handles on two streams {pipe_ms:.3f} ms per search "
          f"({sync_ms / pipe_ms:.2f} x); results identical: {ok}; nomination {hs[0].last_stats()['nomination']}")
    for h in hs: h.close()

"""Does a small kernel get onto a CU whose LDS / registers a filter workgroup holds?  Stream 1: the 2032-query search over
18M rows (filter workgroups: 128 KiB of LDS, 2 x 224 VGPRs per SIMD, one per CU, ~1-4 ms each in the late rounds); stream 2:
small merge launches (merge_topk_device of 8 x [2032, 80] lists: <= 82 VGPRs, no LDS to speak of), each bracketed by
events.  Their durations with stream 1 idle and with stream 1 busy."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP, merge_topk_device

dev = torch.device("cuda:0")
n, nq, k = 18_000_000, 2032, 80
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((2_000_000, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128); ix.adopt_device(xb)
Dp = torch.randn((8, nq, k), generator=g, device=dev).sort(dim=2, descending=True).values.contiguous()
Ip = torch.randint(0, n, (8, nq, k), generator=g, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

def merges(count):
    evs = []
    with torch.cuda.stream(s2):
        for _ in range(count):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s2); merge_topk_device(Dp, Ip); b.record(s2); evs.append((a, b))
    return evs

for _ in range(3): ix.search_device(xq, k); merge_topk_device(Dp, Ip)
torch.cuda.synchronize()
alone = merges(40); torch.cuda.synchronize()
print("merge launches alone: median %.1f us, max %.1f us" % tuple(f(sorted(a.elapsed_time(b) * 1e3 for a, b in alone)) for f in (lambda v: v[len(v) // 2], max)))
import ctypes
from proqa_amd import _lib
from proqa_amd.index import _torch_dtype_code
lib = _lib.load()
D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(s1)
_lib.check(lib.proqa_index_search_begin_device(ix._h, xq.data_ptr(), nq, _torch_dtype_code(xq), k, 0, D.data_ptr(), I.data_ptr(),
                                               status.data_ptr(), s1.cuda_stream))      # enqueued only: ~7.5 ms of GPU work
t1.record(s1)
busy = merges(150)
r = ctypes.c_int(); _lib.check(lib.proqa_index_search_finish(ix._h, ctypes.byref(r)))
torch.cuda.synchronize()
v = sorted(a.elapsed_time(b) * 1e3 for a, b in busy)
print("one search on stream 1: %.2f ms;  150 merge launches enqueued beside it: median %.1f us, 90%% %.1f us, max %.1f us, all together %.2f ms "
      "(first starts %.2f ms after the search, last ends %.2f ms after its start)"
      % (t0.elapsed_time(t1), v[len(v) // 2], v[int(len(v) * 0.9)], v[-1], busy[0][0].elapsed_time(busy[-1][1]),
         t0.elapsed_time(busy[0][0]), t0.elapsed_time(busy[-1][1])))
per = [(t0.elapsed_time(a), a.elapsed_time(b) * 1e3) for a, b in busy]
print("start offset ms -> duration us:", " ".join("%.2f:%.0f" % p for p in per[::6]))

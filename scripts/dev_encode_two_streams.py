"""Round-5 review, next #1(a): two half-batches on two streams (two proqa_encoder handles over the same weights, one workspace
each), free-running, against the single-stream forward of the whole batch (dev; run on the MI355X).

    python scripts/dev_encode_two_streams.py [batch=512] [seq=128] [steps=20]

Prints passages/s of: one stream x B; two streams x B/2 (the second enqueued behind the first by the host: ~half a layer of
stagger); two streams x B (two whole batches in flight); and whether the split output equals the single-stream one bit for bit."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from proqa_amd import _lib  # noqa: E402
from proqa_amd.retriever import BertForRetriever, BERT_BASE, random_state_dict, config_from_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
cfg = config_from_dict(BERT_BASE)
model = BertForRetriever(BERT_BASE, device=dev)
model.load_state_dict({k: v.half().float() for k, v in random_state_dict(cfg, seed=0).items()})
tw = model.towers[False]
h1 = tw._handle
tw._create_encoder(cfg, dev)
h2 = tw._handle
tw._handle = h1
names = model.gemm_kernels()
print("pinned GEMM kernel:", names[False] or "(rocblas_gemm_ex)")
if not names[False]:
    # first forward pins it
    pass
lib = _lib.load()
g = torch.Generator().manual_seed(0)
ids = torch.randint(1000, 30522, (2 * B, S), generator=g)
ids[:, 0], ids[:, -1] = 101, 102
ids = ids.to(dev)
lens = torch.full((2 * B,), S, dtype=torch.int32, device=dev)
flags = _lib.ENC_CLS_ONLY_LAST | _lib.ENC_PACKED


def fwd(handle, lo, hi, out, stream):
    _lib.check(lib.proqa_encoder_forward(handle, ids[lo:hi].data_ptr(), lens[lo:hi].data_ptr(), hi - lo, S, (hi - lo) * S, flags,
                                         out[lo:hi].data_ptr(), 0, stream.cuda_stream))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out_a = torch.zeros((2 * B, 128), dtype=torch.float16, device=dev)
out_b = torch.zeros((2 * B, 128), dtype=torch.float16, device=dev)
# warm both handles at both sizes
for h in (h1, h2):
    fwd(h, 0, B, out_a, s1)
    fwd(h, 0, B // 2, out_a, s1)
torch.cuda.synchronize()
print("pinned GEMM kernel after warm-up:", model.gemm_kernels()[False] or "(rocblas_gemm_ex)")
if not model.gemm_kernels()[False]:
    print("no pinned kernel: the library GEMMs may be stream-K; not running them concurrently")
    sys.exit(0)


def timed(fn, n=STEPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def single():
    fwd(h1, 0, B, out_a, s1)


def split():
    fwd(h1, 0, B // 2, out_b, s1)
    fwd(h2, B // 2, B, out_b, s2)


def two_whole():
    fwd(h1, 0, B, out_b, s1)
    fwd(h2, B, 2 * B, out_b, s2)


def single_half():
    fwd(h1, 0, B // 2, out_a, s1)


for rep in range(3):
    t_single = timed(single)
    t_split = timed(split)
    t_two = timed(two_whole)
    t_half = timed(single_half)
    print(f"rep {rep}: one stream x {B}: {B / t_single:9.0f} passages/s ({t_single * 1e3:.3f} ms)   two streams x {B // 2}: "
          f"{B / t_split:9.0f} ({t_split * 1e3:.3f} ms)   two streams x {B}: {2 * B / t_two:9.0f} ({t_two * 1e3:.3f} ms)   "
          f"one stream x {B // 2}: {B / 2 / t_half:9.0f} ({t_half * 1e3:.3f} ms)")
single()
split()
torch.cuda.synchronize()
print("split output == single-stream output, bit for bit:", bool(torch.equal(out_a[:B], out_b[:B])))

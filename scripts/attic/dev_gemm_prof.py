"""A few launches of the hand-written GEMM (and torch.mm) on one encoder shape, for rocprofv3 passes."""
import sys
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 3072, int(sys.argv[2]) if len(sys.argv) > 2 else 768
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half(); b = torch.randn(N, device=dev).half()
y = torch.empty((M, N), dtype=torch.float16, device=dev)
for _ in range(6):
    _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, epi, _lib.current_stream_ptr()))
for _ in range(6):
    torch.mm(x, w.t())
torch.cuda.synchronize()

"""Hand-written dense layer (proqa_gemm_tn_f16) vs torch.mm (hipBLASLt) on the encoder's shapes: parity and time."""
import sys, time
import torch
sys.path.insert(0, ".")
from proqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)

def own(x, w, b, epi):
    y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float16, device=dev)
    _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                                     x.shape[0], w.shape[0], x.shape[1], epi, _lib.current_stream_ptr()))
    return y

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

# parity on a small problem, every epilogue
for (M, N, K) in [(256, 256, 64), (512, 768, 768), (2048, 3072, 768), (1024, 768, 3072)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.05).half(); b = torch.randn(N, device=dev).half()
    ref = x.float() @ w.float().t()
    for epi in (0, 1, 2):
        r = ref if epi == 0 else ref + b.float()
        if epi == 2: r = torch.nn.functional.gelu(r)
        y = own(x, w, b, epi).float()
        err = (y - r).abs().max().item(); rel = err / r.abs().max().item()
        print(f"parity M={M} N={N} K={K} epi={epi}: max abs err {err:.4g} (rel to max {rel:.3g})")

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for (N, K, name) in [(2304, 768, "qkv"), (768, 768, "attn-out"), (3072, 768, "ffn1"), (768, 3072, "ffn2")]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half(); b = torch.randn(N, device=dev).half()
    fl = 2.0 * M * N * K
    t_lib = timeit(lambda: torch.mm(x, w.t()))
    t0 = timeit(lambda: own(x, w, b, 0))
    t1 = timeit(lambda: own(x, w, b, 1))
    t2 = timeit(lambda: own(x, w, b, 2))
    print(f"{name:8s} M={M} N={N} K={K}: hipBLASLt {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.0f} TF | own {t0*1e6:7.1f} us {fl/t0/1e12:6.0f} TF | "
          f"+bias {t1*1e6:7.1f} us | +bias+gelu {t2*1e6:7.1f} us")

"""Probe: rocblas_gemm_flags_use_cu_efficiency (0x2) vs default at the encoder's GEMM shapes (interleaved)."""
import ctypes, os, time, statistics
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"), mode=ctypes.RTLD_GLOBAL)
h = ctypes.c_void_p(); assert lib.rocblas_create_handle(ctypes.byref(h)) == 0
V, I = ctypes.c_void_p, ctypes.c_int
lib.rocblas_gemm_ex.argtypes = [V, I, I, I, I, I, V, V, I, I, V, I, I, V, V, I, I, V, I, I, I, I, ctypes.c_int32, ctypes.c_uint32]
lib.rocblas_set_stream.argtypes = [V, V]
lib.rocblas_set_stream(h, V(torch.cuda.current_stream().cuda_stream))
dev = torch.device("cuda", 0)
alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
def run(x, w, out, flags):
    M, K = x.shape; N = w.shape[0]
    return lib.rocblas_gemm_ex(h, 112, 111, N, M, K, ctypes.byref(alpha), w.data_ptr(), 150, K, x.data_ptr(), 150, K,
                               ctypes.byref(beta), out.data_ptr(), 150, N, out.data_ptr(), 150, N, 151, 0, 0, flags)
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K) in [(65536, 2304, 768), (65536, 768, 768), (65536, 3072, 768), (65536, 768, 3072), (41216, 3072, 768), (41216, 768, 3072)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half()
    out = torch.empty((M, N), device=dev, dtype=torch.float16)
    t = {0: [], 2: []}
    for r in range(8):
        for f in (0, 2):
            t[f].append(bench(lambda: run(x, w, out, f)))
    print(M, N, K, {f: round(statistics.median(v) * 1e3, 4) for f, v in t.items()})

"""How the default rocBLAS pick behaves vs the row count M (packed batches have arbitrary M)."""
import ctypes, os, time
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"), mode=ctypes.RTLD_GLOBAL)
h = ctypes.c_void_p(); assert lib.rocblas_create_handle(ctypes.byref(h)) == 0
V, I = ctypes.c_void_p, ctypes.c_int
lib.rocblas_gemm_ex.argtypes = [V, I, I, I, I, I, V, V, I, I, V, I, I, V, V, I, I, V, I, I, I, I, ctypes.c_int32, ctypes.c_uint32]
lib.rocblas_set_stream.argtypes = [V, V]
lib.rocblas_set_stream(h, V(torch.cuda.current_stream().cuda_stream))
dev = torch.device("cuda", 0)
alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
def run(x, w, out):
    M, K = x.shape; N = w.shape[0]
    lib.rocblas_gemm_ex(h, 112, 111, N, M, K, ctypes.byref(alpha), w.data_ptr(), 150, K, x.data_ptr(), 150, K,
                        ctypes.byref(beta), out.data_ptr(), 150, N, out.data_ptr(), 150, N, 151, 0, 0, 0)
for (N, K) in [(3072, 768), (768, 3072), (2304, 768)]:
    w = (torch.randn((N, K), device=dev) * 0.02).half()
    line = []
    import sys
    Ms = [int(a) for a in sys.argv[1:]] or [32768, 36864, 40960, 41216, 41472, 43008, 45056, 49152, 53248, 57344, 61440, 65536]
    for M in Ms:
        x = torch.randn((M, K), device=dev).half(); out = torch.empty((M, N), device=dev, dtype=torch.float16)
        for _ in range(3): run(x, w, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run(x, w, out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        line.append(f"{M}:{2*M*N*K/dt/1e12:.0f}")
    print(f"N={N} K={K} TF/s by M  " + "  ".join(line))

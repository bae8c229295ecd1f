"""Feasibility probe: call torch's bundled rocBLAS (rocblas_gemm_ex) through ctypes on torch tensors and
compare result/time with torch.mm at the encoder's GEMM shapes."""
import ctypes, os, time
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"), mode=ctypes.RTLD_GLOBAL)
h = ctypes.c_void_p()
assert lib.rocblas_create_handle(ctypes.byref(h)) == 0
R_F16, R_F32 = 150, 151            # rocblas_datatype_f16_r, rocblas_datatype_f32_r
OP_N, OP_T = 111, 112              # rocblas_operation_none / transpose
lib.rocblas_gemm_ex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int32, ctypes.c_uint32]
lib.rocblas_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
st = torch.cuda.current_stream().cuda_stream
assert lib.rocblas_set_stream(h, ctypes.c_void_p(st)) == 0

def gemm(x, w, out):       # out[M,N] = x[M,K] @ w[N,K]^T  (row-major) == col-major: out'[N,M] = w'^T[N,K] x'[K,M]
    M, K = x.shape; N = w.shape[0]
    rc = lib.rocblas_gemm_ex(h, OP_T, OP_N, N, M, K, ctypes.byref(alpha), w.data_ptr(), R_F16, K, x.data_ptr(), R_F16, K,
                             ctypes.byref(beta), out.data_ptr(), R_F16, N, out.data_ptr(), R_F16, N, R_F32, 0, 0, 0)
    assert rc == 0, rc

for (M, N, K) in [(65536, 2304, 768), (65536, 768, 768), (65536, 3072, 768), (65536, 768, 3072), (512, 768, 768), (41000, 3072, 768)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half()
    o1 = torch.empty((M, N), device=dev, dtype=torch.float16); o2 = torch.empty_like(o1)
    gemm(x, w, o1); torch.mm(x, w.t(), out=o2); torch.cuda.synchronize()
    err = (o1.float() - o2.float()).abs().max().item()
    for fn, name in ((lambda: gemm(x, w, o1), "rocblas"), (lambda: torch.mm(x, w.t(), out=o2), "torch.mm")):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"M={M} N={N} K={K} {name}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.0f} TF/s  (max diff {err:.3g})")

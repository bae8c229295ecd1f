"""Probe: does choosing among rocBLAS' solutions beat its default pick at the encoder's GEMM shapes?
Times every solution rocblas_gemm_ex_get_solutions lists (sustained: 20 calls each) vs the default."""
import ctypes, os, time
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"), mode=ctypes.RTLD_GLOBAL)
h = ctypes.c_void_p(); assert lib.rocblas_create_handle(ctypes.byref(h)) == 0
R_F16, R_F32, OP_N, OP_T = 150, 151, 111, 112
V, I = ctypes.c_void_p, ctypes.c_int
lib.rocblas_gemm_ex.argtypes = [V, I, I, I, I, I, V, V, I, I, V, I, I, V, V, I, I, V, I, I, I, I, ctypes.c_int32, ctypes.c_uint32]
lib.rocblas_gemm_ex_get_solutions.argtypes = [V, I, I, I, I, I, V, V, I, I, V, I, I, V, V, I, I, V, I, I, I, I, ctypes.c_uint32,
                                               ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
dev = torch.device("cuda", 0)
alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
lib.rocblas_set_stream.argtypes = [V, V]
lib.rocblas_set_stream(h, V(torch.cuda.current_stream().cuda_stream))

def run(x, w, out, algo, sol):
    M, K = x.shape; N = w.shape[0]
    return lib.rocblas_gemm_ex(h, OP_T, OP_N, N, M, K, ctypes.byref(alpha), w.data_ptr(), R_F16, K, x.data_ptr(), R_F16, K,
                               ctypes.byref(beta), out.data_ptr(), R_F16, N, out.data_ptr(), R_F16, N, R_F32, algo, sol, 0)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

tot_def = tot_best = 0.0
for (M, N, K) in [(65536, 2304, 768), (65536, 768, 768), (65536, 3072, 768), (65536, 768, 3072)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half()
    out = torch.empty((M, N), device=dev, dtype=torch.float16)
    n = ctypes.c_int(0)
    rc = lib.rocblas_gemm_ex_get_solutions(h, OP_T, OP_N, N, M, K, ctypes.byref(alpha), w.data_ptr(), R_F16, K, x.data_ptr(), R_F16, K,
                                           ctypes.byref(beta), out.data_ptr(), R_F16, N, out.data_ptr(), R_F16, N, R_F32, 0, 0, None, ctypes.byref(n))
    arr = (ctypes.c_int * max(n.value, 1))()
    lib.rocblas_gemm_ex_get_solutions(h, OP_T, OP_N, N, M, K, ctypes.byref(alpha), w.data_ptr(), R_F16, K, x.data_ptr(), R_F16, K,
                                      ctypes.byref(beta), out.data_ptr(), R_F16, N, out.data_ptr(), R_F16, N, R_F32, 0, 0, arr, ctypes.byref(n))
    t_def = bench(lambda: run(x, w, out, 0, 0))
    res = []
    for s in list(arr)[:n.value]:
        if run(x, w, out, 1, s) != 0:
            continue
        res.append((bench(lambda: run(x, w, out, 1, s), 8), s))
    res.sort()
    t_best = bench(lambda: run(x, w, out, 1, res[0][1])) if res else t_def
    print(f"M={M} N={N} K={K}: rc={rc} {n.value} solutions; default {t_def*1e3:.3f} ms; best {t_best*1e3:.3f} ms (sol {res[0][1] if res else None}); top3 {[(round(t*1e3,3), s) for t, s in res[:3]]}")
    tot_def += t_def; tot_best += min(t_best, t_def)
print(f"sum over the four shapes: default {tot_def*1e3:.3f} ms, tuned {tot_best*1e3:.3f} ms ({(1-tot_best/tot_def)*100:.1f} % less)")

print("interleaved A/B (10 rounds x 10 calls), ms:")
import statistics
for (M, N, K) in [(65536, 2304, 768), (65536, 768, 768), (65536, 3072, 768), (65536, 768, 3072)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half()
    out = torch.empty((M, N), device=dev, dtype=torch.float16)
    cands = {"default": (0, 0), "624961": (1, 624961), "625109": (1, 625109), "625108": (1, 625108)}
    times = {k: [] for k in cands}
    for r in range(10):
        for name, (algo, sol) in cands.items():
            times[name].append(bench(lambda: run(x, w, out, algo, sol), 10))
    print(M, N, K, {k: round(statistics.median(v) * 1e3, 4) for k, v in times.items()})

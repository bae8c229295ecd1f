// Internal interface of lt_gemm.cpp: the encoder's library GEMMs on a hipBLASLt kernel pinned by name (see there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace proqa {

struct LtGemm;
// nullptr when the process holds no hipBLASLt with the extension API (the caller stays on rocblas_gemm_ex)
LtGemm* lt_gemm_create();
void lt_gemm_destroy(LtGemm* g);
// out[M,N] = x[M,K] . w[N,K]^T, row-major fp16, fp32 accumulate, on `st`.  0 = launched; 1 = no pinned kernel for this
// library / problem (the caller falls back); < 0 = error (proqa_last_error)
int lt_gemm_tn(LtGemm* g, const void* x, const void* w, void* out, int64_t M, int N, int K, hipStream_t st);
// name of the pinned kernel ("" before the first large product, or when none of the preferred names exists)
const char* lt_gemm_kernel_name(const LtGemm* g);

}  // namespace proqa

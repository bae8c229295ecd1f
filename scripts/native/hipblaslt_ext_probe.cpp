// Dev probe: EVERY fp16 TN algorithm of the hipBLASLt build PyTorch bundles (hipblaslt_ext::getAllAlgos), timed on the
// encoder's dense shapes at M = 65536 with random operands, with its kernel name -- what a deterministic by-name choice at
// proqa_encoder_create would pick from.
//   hipcc --offload-arch=gfx950 -O2 scripts/native/hipblaslt_ext_probe.cpp -L<torch>/lib -lhipblaslt -o gpurun_build/hipblaslt_ext_probe
//   LD_LIBRARY_PATH=<torch>/lib gpurun_build/hipblaslt_ext_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { auto _s = (x); if (_s != 0) { printf("%s failed: %d\n", #x, (int)_s); return 1; } } while (0)
int main() {
  const int64_t M = 65536;
  const int shapes[4][2] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  void* ws; const size_t ws_bytes = 128 << 20; CK(hipMalloc(&ws, ws_bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<hipblasLtMatmulHeuristicResult_t> all;
  CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N, HIP_R_16F, HIP_R_16F, HIP_R_16F,
                                HIP_R_16F, HIPBLAS_COMPUTE_32F, all));
  printf("%zu algorithms in all\n", all.size());
  for (auto& s : shapes) {
    const int N = s[0], K = s[1];
    void *x, *w, *out; CK(hipMalloc(&x, M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&out, M * N * 2));
    {
      std::vector<_Float16> hx((size_t)M * K), hw((size_t)N * K);
      unsigned long long z = 88172645463325252ull;
      auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (float)((z >> 11) & 0xFFFF) / 32768.0f - 1.0f; };
      for (auto& v : hx) v = (_Float16)rnd();
      for (auto& v : hw) v = (_Float16)(0.05f * rnd());
      CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, K, N, K));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, K, M, K));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16F, N, M, N));
    const float alpha = 1.f, beta = 0.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Row { float us; int index; std::string name; };
    std::vector<Row> rows;
    // the heuristic's own first choice for reference
    {
      hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
      uint64_t maxws = ws_bytes; CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &maxws, sizeof maxws));
      hipblasLtMatmulHeuristicResult_t r1; int got = 0;
      CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 1, &r1, &got));
      if (got) printf("N=%d K=%d: heuristic first choice = index %d  %s\n", N, K, hipblaslt_ext::getIndexFromAlgo(r1.algo),
                      hipblaslt_ext::getKernelNameFromAlgo(h, r1.algo).c_str());
    }
    int tried = 0;
    for (auto& r : all) {
      size_t need = 0;
      if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb, &beta, lc, lc, r.algo, need) != HIPBLAS_STATUS_SUCCESS || need > ws_bytes) continue;
      ++tried;
      bool ok = true;
      for (int it = 0; it < 2 && ok; ++it)
        ok = hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, out, lc, out, lc, &r.algo, ws, ws_bytes, st) == HIPBLAS_STATUS_SUCCESS;
      if (!ok) continue;
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < 6; ++it) hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, out, lc, out, lc, &r.algo, ws, ws_bytes, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const float us = ms / 6 * 1000;
      if (us < 2000.f) rows.push_back({us, hipblaslt_ext::getIndexFromAlgo(r.algo), hipblaslt_ext::getKernelNameFromAlgo(h, r.algo)});
    }
    std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.us < b.us; });
    // re-time the best eight properly (30 launches each, interleaved twice)
    printf("N=%d K=%d: %d supported, %zu timed; best:\n", N, K, tried, rows.size());
    for (size_t j = 0; j < rows.size() && j < 5; ++j) printf("  %.1f us  index %d  %s\n", rows[j].us, rows[j].index, rows[j].name.c_str());
    hipFree(x); hipFree(w); hipFree(out);
  }
  return 0;
}

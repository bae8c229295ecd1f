// Dev probe: the algorithms hipblasLtMatmulAlgoGetHeuristic offers for the encoder's dense shapes, each timed.
//   hipcc --offload-arch=gfx950 -O2 scripts/native/hipblaslt_probe.cpp -lhipblaslt -o gpurun_build/hipblaslt_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto _s = (x); if (_s != 0) { printf("%s failed: %d\n", #x, (int)_s); return 1; } } while (0)
int main(int argc, char** argv) {
  const int64_t M = 65536;
  const int shapes[3][2] = {{2304, 768}, {768, 768}, {768, 3072}};
  const int want = argc > 1 ? atoi(argv[1]) : 64;
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  void* ws; const size_t ws_bytes = 64 << 20; CK(hipMalloc(&ws, ws_bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (auto& s : shapes) {
    const int N = s[0], K = s[1];
    void *x, *w, *out; CK(hipMalloc(&x, M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&out, M * N * 2));
    {   // random operands: the chip is power-limited, constant data would run ~30 % faster than real activations
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
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, K, N, K));   // weight [N,K] row-major = K x N column-major
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, K, M, K));   // tokens [M,K] row-major = K x M column-major
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16F, N, M, N));   // out [M,N] row-major = N x M column-major
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    uint64_t maxws = ws_bytes; CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &maxws, sizeof maxws));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
    int got = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, want, res.data(), &got));
    printf("N=%d K=%d: %d algorithms\n", N, K, got);
    const float alpha = 1.f, beta = 0.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::pair<float, int>> times;
    for (int i = 0; i < got; ++i) {
      if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > ws_bytes) continue;
      bool ok = true;
      for (int r = 0; r < 3 && ok; ++r)
        ok = hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, out, lc, out, lc, &res[i].algo, ws, ws_bytes, st) == HIPBLAS_STATUS_SUCCESS;
      if (!ok) continue;
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < 30; ++r) hipblasLtMatmul(h, desc, &alpha, w, la, x, lb, &beta, out, lc, out, lc, &res[i].algo, ws, ws_bytes, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      times.push_back({ms / 30 * 1000, i});
    }
    if (!times.empty()) printf("  heuristic's first choice: %.1f us\n", times[0].second == 0 ? times[0].first : -1.f);
    std::sort(times.begin(), times.end());
    for (size_t j = 0; j < times.size() && j < 6; ++j)
      printf("  rank %zu: %.1f us  (heuristic position %d, algo index %d, workspace %zu)\n", j, times[j].first, times[j].second,
             *(int*)res[times[j].second].algo.data, res[times[j].second].workspaceSize);
    hipFree(x); hipFree(w); hipFree(out);
  }
  return 0;
}

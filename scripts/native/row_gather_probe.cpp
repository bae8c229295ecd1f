// Probe: the rate at which a merge-shaped launch (2032 workgroups of 4 waves, ~264 random rows each) gathers 256-byte fp16
// rows out of a 4.6 GB array, in the access pattern topk_merge<NOMINATED_I8> uses (a wave takes 32 rows as the A operand of
// v_mfma_f32_32x32x16_f16: lane (row = lane & 31, half = lane >> 5) loads the 16-byte pieces 2j + half, j = 0..7 -- every
// load instruction touches 32 rows, 32 bytes of each) against a line-coalesced pattern (16 lanes per row and instruction:
// four whole rows per instruction).  Question: is the re-scoring gather bound by requests or by bytes?  (ABLATIONS R5.10)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe scripts/native/row_gather_probe.cpp && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ rows, const unsigned* __restrict__ ids, int per_wg,
                                              unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned* my = ids + (size_t)blockIdx.x * per_wg;
  u32x4 acc = {0, 0, 0, 0};
  for (int b = wave * 32; b < per_wg; b += 4 * 32) {
    u32x4 v[8];
    if (PATTERN == 0) {          // the merge's pattern: lane -> (row b + (lane & 31), pieces 2j + (lane >> 5))
      const int r = b + (lane & 31);
      const char* p = rows + (size_t)my[r < per_wg ? r : per_wg - 1] * 256 + (lane >> 5) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const u32x4*)(p + j * 32);
    } else {                     // line-coalesced: instruction j covers rows b + 4j .. b + 4j + 3, 16 lanes x 16 B each
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = b + 4 * j + (lane >> 4);
        v[j] = *(const u32x4*)(rows + (size_t)my[r < per_wg ? r : per_wg - 1] * 256 + (lane & 15) * 16);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = 1;   // (keeps the loads)
}

int main(int argc, char** argv) {
  const long long n_rows = argc > 1 ? atoll(argv[1]) : 18000000;
  const int n_wg = 2032;
  char* rows;
  unsigned *ids, *out;
  CK(hipMalloc(&rows, (size_t)n_rows * 256));
  CK(hipMemset(rows, 1, (size_t)n_rows * 256));
  CK(hipMalloc(&out, n_wg * 4));
  for (int per_wg : {64, 128, 264, 512, 2048}) {
    std::vector<unsigned> h((size_t)n_wg * per_wg);
    unsigned long long s = 88172645463325252ull;
    for (auto& x : h) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      x = (unsigned)(s % (unsigned long long)n_rows);
    }
    CK(hipMalloc(&ids, h.size() * 4));
    CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int pattern = 0; pattern < 2; ++pattern) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (pattern == 0) hipLaunchKernelGGL(gather<0>, dim3(n_wg), dim3(256), 0, 0, rows, ids, per_wg, out);
        else hipLaunchKernelGGL(gather<1>, dim3(n_wg), dim3(256), 0, 0, rows, ids, per_wg, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      const double bytes = (double)n_wg * per_wg * 256;
      printf("%4d rows per workgroup (%6.1f MB), %s: %7.1f us  %6.2f TB/s\n", per_wg, bytes / 1e6,
             pattern == 0 ? "MFMA-fragment pattern (32 rows x 32 B per load)" : "line-coalesced (4 rows x 256 B per load)     ", best * 1e3,
             bytes / (best * 1e-3) / 1e12);
    }
    CK(hipFree(ids));
  }
  return 0;
}

// Dev probe: issue cost of a 64-bit compare-exchange against a (hi, lo) 32-bit one on gfx950 (one wave per SIMD).
//   hipcc --offload-arch=gfx950 -O3 scripts/native/cmp_rate_probe.cpp -o gpurun_build/cmp_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void probe(unsigned long long* out, unsigned long long seed, int iters) {
  unsigned long long v[16];
  for (int j = 0; j < 16; ++j) v[j] = (seed * (threadIdx.x + 1) * (j + 3) ^ (seed >> (j + 1))) & 0x7FEFFFFFFFFFFFFFull;   // finite doubles
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if ((j & s) == 0) {
          const unsigned long long x = v[j], y = v[j + s];
          bool sw;
          if (MODE == 0) {
            sw = x < y;
          } else if (MODE == 2) {
            sw = __longlong_as_double((long long)x) < __longlong_as_double((long long)y);
          } else if (MODE == 3) {
            sw = (unsigned)(x >> 32) < (unsigned)(y >> 32);
          } else {
            const unsigned xh = (unsigned)(x >> 32), yh = (unsigned)(y >> 32), xl = (unsigned)x, yl = (unsigned)y;
            sw = (xh < yh) | ((xh == yh) & (xl < yl));
          }
          v[j] = sw ? y : x;
          v[j + s] = sw ? x : y;
        }
      }
    }
    v[0] += it;   // keep the loop from collapsing
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long acc = 0;
  for (int j = 0; j < 16; ++j) acc ^= v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, ((1 << 20) + 8) * 8);
  const int iters = 2000;
  for (int mode = 0; mode < 4; ++mode) {
    for (int waves = 1; waves <= 2; ++waves) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256 * waves), 0, 0, d, 0x9E3779B97F4A7C15ull, iters);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256 * waves), 0, 0, d, 0x9E3779B97F4A7C15ull, iters);
        else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256 * waves), 0, 0, d, 0x9E3779B97F4A7C15ull, iters);
        else hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256 * waves), 0, 0, d, 0x9E3779B97F4A7C15ull, iters);
        hipDeviceSynchronize();
      }
      unsigned long long t; hipMemcpy(&t, d + (1 << 20), 8, hipMemcpyDeviceToHost);
      printf("%s compare, %d wave(s) per SIMD: %.1f ticks per compare-exchange (32 per iteration)\n", mode == 0 ? "u64" : mode == 1 ? "hi/lo u32" : mode == 2 ? "f64" : "hi32 only", waves,
             (double)t / iters / 32.0);
    }
  }
  return 0;
}

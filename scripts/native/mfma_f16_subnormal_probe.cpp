// Probe: does v_mfma_f32_32x32x16_f16 treat fp16 SUBNORMAL operands (|v| < 2^-14) as their values or flush them to zero?
// (the error margins of the exact-float32 mode and of the int8 nomination scan carry a term for the second case)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe scripts/native/mfma_f16_subnormal_probe.cpp && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const _Float16* A, const _Float16* B, float* out) {
  const int lane = threadIdx.x, li = lane & 31, half = lane >> 5;
  f32x16 acc = {0};
  for (int j = 0; j < 8; ++j) {
    const f16x8 a = *(const f16x8*)(A + li * 128 + (2 * j + half) * 8);
    const f16x8 b = *(const f16x8*)(B + li * 128 + (2 * j + half) * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
}
int main() {
  static _Float16 hA[32 * 128], hB[32 * 128];
  // row i of A: subnormal values (i + 1) * 2^-24 (the smallest subnormal is 2^-24); column j of B: normal values (1 + j) * 64
  for (int i = 0; i < 32; ++i)
    for (int k = 0; k < 128; ++k) {
      hA[i * 128 + k] = (_Float16)((float)(i + 1) * 5.9604644775390625e-08f);
      hB[i * 128 + k] = (_Float16)((float)(1 + i) * 64.0f);
    }
  _Float16 *dA, *dB;
  float* dO;
  (void)hipMalloc(&dA, sizeof hA);
  (void)hipMalloc(&dB, sizeof hB);
  (void)hipMalloc(&dO, 32 * 32 * 4);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dO);
  static float hO[32 * 32];
  (void)hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
  int bad = 0, zero = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const double ref = 128.0 * (double)(float)hA[i * 128] * (double)(float)hB[j * 128];
      if (hO[i * 32 + j] == 0.0f) ++zero;
      if ((double)hO[i * 32 + j] != ref) ++bad;
    }
  printf("subnormal A x normal B: %d of 1024 results differ from the exact product sum, %d are zero (flushed); sample got %g want %g\n", bad,
         zero, hO[5 * 32 + 3], 128.0 * (double)(float)hA[5 * 128] * (double)(float)hB[3 * 128]);
  // and the mirrored case: subnormal B
  (void)hipMemcpy(dA, hB, sizeof hA, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hA, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dO);
  (void)hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
  bad = zero = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const double ref = 128.0 * (double)(float)hB[i * 128] * (double)(float)hA[j * 128];
      if (hO[i * 32 + j] == 0.0f) ++zero;
      if ((double)hO[i * 32 + j] != ref) ++bad;
    }
  printf("normal A x subnormal B: %d of 1024 results differ, %d are zero\n", bad, zero);
  return 0;
}

// Probe: does v_mfma_i32_32x32x32_i8 pair the bytes of the A and B operands symmetrically, i.e. is a dot product invariant
// under "piece 2j + half of row li at k-step j" on both operands (the convention of mips_filter_i8)?  Also prints the
// register -> row map of the result.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/mfma_i8_layout_probe scripts/native/mfma_i8_layout_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const signed char* A, const signed char* B, int* out) {
  const int lane = threadIdx.x, li = lane & 31, half = lane >> 5;
  i32x16 acc = {0};
  for (int j = 0; j < 4; ++j) {
    const i32x4 a = *(const i32x4*)(A + li * 128 + (2 * j + half) * 16);
    const i32x4 b = *(const i32x4*)(B + li * 128 + (2 * j + half) * 16);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
}
int main() {
  signed char hA[32 * 128], hB[32 * 128];
  srand(1);
  for (int i = 0; i < 32 * 128; ++i) {
    hA[i] = (signed char)(rand() % 255 - 127);
    hB[i] = (signed char)(rand() % 255 - 127);
  }
  signed char *dA, *dB;
  int* dO;
  hipMalloc(&dA, sizeof hA);
  hipMalloc(&dB, sizeof hB);
  hipMalloc(&dO, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dO);
  int hO[32 * 32];
  hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int ref = 0;
      for (int k = 0; k < 128; ++k) ref += (int)hA[i * 128 + k] * (int)hB[j * 128 + k];
      if (ref != hO[i * 32 + j]) {
        if (bad < 8) printf("mismatch row %d query %d: got %d want %d\n", i, j, hO[i * 32 + j], ref);
        ++bad;
      }
    }
  printf("mismatches: %d of 1024\n", bad);
  return bad != 0;
}

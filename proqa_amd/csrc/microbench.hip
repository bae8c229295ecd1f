// Measured ceilings of the box the benchmark runs on (bench.py: "peak_measured"): a float4 stream copy and a
// register-resident MFMA loop.  They are what SURVEY.md section 8(d) asks to be recorded next to the spec peaks
// (HBM3E 8 TB/s, 2.5 PFLOP/s dense fp16 MFMA) that the roofline fractions are priced against.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdlib.h>

#include "common.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Each workgroup streams its own contiguous slice (as the search kernel streams a corpus chunk): 16 bytes per lane
// and access, eight accesses in flight per lane.
constexpr int kStreamUnroll = 8;
__global__ __launch_bounds__(256) void stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n_vec) {
  const size_t per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const size_t hi = lo + per_wg < n_vec ? lo + per_wg : n_vec;
  size_t i = lo + threadIdx.x;
  for (; i + (kStreamUnroll - 1) * 256 < hi; i += kStreamUnroll * 256) {
    f32x4 v[kStreamUnroll];
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * 256);
  }
  for (; i < hi; i += 256) dst[i] = src[i];
}

// read-only variant: the search kernel's traffic is all reads
__global__ __launch_bounds__(256) void stream_read(const f32x4* __restrict__ src, float* __restrict__ sink, size_t n_vec) {
  const size_t per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const size_t hi = lo + per_wg < n_vec ? lo + per_wg : n_vec;
  size_t i = lo + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (kStreamUnroll - 1) * 256 < hi; i += kStreamUnroll * 256) {
    f32x4 v[kStreamUnroll];
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < kStreamUnroll; ++u) acc += v[u];
  }
  for (; i < hi; i += 256) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) *sink = acc[0];   // keeps the loads alive
}

// 4 independent 32x32x16 fp16 MFMA chains per wave, operands from registers: nothing but the matrix pipe.
// `zero` selects all-zero operands (the clock the chip sustains depends on the operand bits).
__global__ __launch_bounds__(256) void mfma_loop(int iters, int zero, float* __restrict__ sink) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    const unsigned h = (threadIdx.x * 2654435761u + e * 40503u + blockIdx.x * 97u) >> 7;
    a[e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(h & 1023) / 512.f - 1.f);
    b[e] = zero ? (_Float16)0.f : (_Float16)((float)(int)((h >> 10) & 1023) / 512.f - 1.f);
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  const f32x16 s = c0 + c1 + c2 + c3;
  float t = 0.f;
  for (int e = 0; e < 16; ++e) t += s[e];
  if (t == 1.2345e-30f) *sink = t;
}

// the same loop on the int8 matrix instruction (v_mfma_i32_32x32x32_i8: twice the k of the fp16 form per issue)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop_i8(int iters, int zero, int* __restrict__ sink) {
  i32x4 a, b;
  for (int e = 0; e < 4; ++e) {
    const unsigned h = (threadIdx.x * 2654435761u + e * 40503u + blockIdx.x * 97u);
    a[e] = zero ? 0 : (int)(h * 2246822519u);          // four random int8 per register
    b[e] = zero ? 0 : (int)(h * 3266489917u + 12345u);
  }
  i32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
  }
  const i32x16 s = c0 + c1 + c2 + c3;
  int t = 0;
  for (int e = 0; e < 16; ++e) t += s[e];
  if (t == 0x12345678) *sink = t;
}

// the 16x16x64 form of the int8 instruction (v_mfma_i32_16x16x64_i8: 8 passes, the shape MI355X_MICROARCH.md quotes the
// >= 3944 TOPS ceiling for): 4 chains of 4 accumulator registers each
__global__ __launch_bounds__(256) void mfma_loop_i8_16x16x64(int iters, int zero, int* __restrict__ sink) {
  i32x4 a, b;
  for (int e = 0; e < 4; ++e) {
    const unsigned h = (threadIdx.x * 2654435761u + e * 40503u + blockIdx.x * 97u);
    a[e] = zero ? 0 : (int)(h * 2246822519u);
    b[e] = zero ? 0 : (int)(h * 3266489917u + 12345u);
  }
  i32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
  }
  const i32x4 s = c0 + c1 + c2 + c3;
  const int t = s[0] + s[1] + s[2] + s[3];
  if (t == 0x12345678) *sink = t;
}

// What does VALU work beside the matrix instructions cost?  The int8 scan's unit is 8 MFMAs (two accumulator chains of 4)
// followed by the examination of the 32 accumulators (~24 plain VALU instructions); here: the same 8 MFMAs followed by NV
// three-operand VALU instructions over the accumulators (two dependency chains, as the scan's two query blocks), no branch, no memory, FOUR
// waves per SIMD (1024 threads per CU) like the scan's two 8-wave workgroups.  The rate over NV = 0, 8, 16, ... separates
// "the other waves' VALU hides under a wave's MFMAs" (flat) from "VALU issue and matrix issue share the SIMD" (falling).
template <int NV>
__global__ __launch_bounds__(512, 2) void mfma_i8_valu_loop(int iters, int* __restrict__ sink) {
  i32x4 a, b;
  for (int e = 0; e < 4; ++e) {
    const unsigned h = (threadIdx.x * 2654435761u + e * 40503u + blockIdx.x * 97u);
    a[e] = (int)(h * 2246822519u);
    b[e] = (int)(h * 3266489917u + 12345u);
  }
  int m0 = 0, m1 = 0;
  for (int i = 0; i < iters; ++i) {
    i32x16 c0 = {0}, c1 = {0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
    }
    if (NV == 0) {   // keep the results alive with one instruction per chain
      m0 ^= c0[0];
      m1 ^= c1[0];
    }
#pragma unroll
    for (int v = 0; v < NV / 2; ++v) {
      const int x0 = c0[(2 * v) & 15], y0 = c0[(2 * v + 1) & 15], x1 = c1[(2 * v) & 15], y1 = c1[(2 * v + 1) & 15];
      // (plain C++: hipcc makes each line ONE three-operand VALU instruction, v_xad_u32, and inserts the MFMA -> VALU wait
      // states itself; a max tree would be folded -- the maximum is idempotent -- whatever NV says)
      m0 = (m0 ^ x0) + y0;
      m1 = (m1 ^ x1) + y1;
    }
  }
  if (m0 + m1 == 0x12345678) *sink = m0;
}

// What a grid-wide barrier costs: a cooperative launch of `grid` 256-thread workgroups that meet at n_syncs barriers
// (cooperative_groups::grid_group::sync, i.e. device-scope release / acquire across the XCDs), touching one cache line each in
// between -- the price list of a fused small-batch encoder (ABLATIONS R6.12).
__global__ __launch_bounds__(256) void grid_sync_loop(int n_syncs, unsigned* __restrict__ buf) {
  cooperative_groups::grid_group g = cooperative_groups::this_grid();
  unsigned v = 0;
  for (int i = 0; i < n_syncs; ++i) {
    if (threadIdx.x == 0) buf[blockIdx.x * 16] = v + (unsigned)i;
    g.sync();
    if (threadIdx.x == 0) v += buf[((blockIdx.x + 1) % gridDim.x) * 16];
  }
  if (threadIdx.x == 0) buf[blockIdx.x * 16 + 1] = v;
}

}  // namespace
}  // namespace proqa

using namespace proqa;

extern "C" {

// kind 0: copy (bytes read + bytes written are both counted), kind 1: read only.  `buf` holds 2*bytes (copy) or
// bytes (read); result in GB/s of the best of `reps` launches.
int proqa_microbench_stream(void* buf, size_t bytes, int kind, int reps, void* stream, double* gbs) {
  if (!buf || !gbs || bytes < (1u << 20) || reps <= 0 || kind < 0 || kind > 1)
    return fail(PROQA_EINVAL, "microbench_stream: bad argument");
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  PROQA_HIP(hipEventCreate(&e0));
  PROQA_HIP(hipEventCreate(&e1));
  const size_t n_vec = bytes / 16;
  const unsigned grid = (unsigned)device_cu_count() * (getenv("PROQA_STREAM_WGS") ? atoi(getenv("PROQA_STREAM_WGS")) : 8);
  float* sink = nullptr;
  PROQA_HIP(hipMalloc((void**)&sink, 4));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {
    PROQA_HIP(hipEventRecord(e0, st));
    if (kind == 0)
      hipLaunchKernelGGL(stream_copy, dim3(grid), dim3(256), 0, st, (const f32x4*)buf, (f32x4*)((char*)buf + bytes), n_vec);
    else
      hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, st, (const f32x4*)buf, sink, n_vec);
    PROQA_HIP(hipEventRecord(e1, st));
    PROQA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PROQA_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;   // first launch warms up
  }
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbs = (kind == 0 ? 2.0 : 1.0) * (double)(n_vec * 16) / (best * 1e-3) / 1e9;
  return PROQA_OK;
}

// dense fp16 MFMA rate of a kernel that does nothing else: `ms_target` sets the launch length (the sustained
// clock depends on it); zero_operands != 0 runs all-zero inputs.  Result in TFLOP/s.
int proqa_microbench_mfma(double ms_target, int zero_operands, void* stream, double* tflops) {
  if (!tflops || !(ms_target > 0)) return fail(PROQA_EINVAL, "microbench_mfma: bad argument");
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  PROQA_HIP(hipEventCreate(&e0));
  PROQA_HIP(hipEventCreate(&e1));
  float* sink = nullptr;
  PROQA_HIP(hipMalloc((void**)&sink, 4));
  const unsigned grid = (unsigned)device_cu_count() * 2;   // 8 waves per CU, 2 per SIMD
  // one MFMA = 32 cycles per SIMD; two waves share a SIMD: iters * 4 * 2 * 32 cycles per launch at ~2.4 GHz
  int iters = (int)(ms_target * 1e-3 * 2.4e9 / (4 * 2 * 32));
  if (iters < 64) iters = 64;
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    PROQA_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, st, iters, zero_operands, sink);
    PROQA_HIP(hipEventRecord(e1, st));
    PROQA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PROQA_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double flops = (double)grid * 4 /*waves*/ * (double)iters * 4 /*chains*/ * 2.0 * 32 * 32 * 16;
  *tflops = flops / (best * 1e-3) / 1e12;
  return PROQA_OK;
}

// the int8 rate of the same kind of loop (the nomination scan's instruction).  Result in TOP/s (2 x MAC).
// shape 0: v_mfma_i32_32x32x32_i8 (what mips_filter_i8 issues), 1: v_mfma_i32_16x16x64_i8.
int proqa_microbench_mfma_i8_shape(double ms_target, int zero_operands, int shape, void* stream, double* tops) {
  if (!tops || !(ms_target > 0) || shape < 0 || shape > 1) return fail(PROQA_EINVAL, "microbench_mfma_i8: bad argument");
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  PROQA_HIP(hipEventCreate(&e0));
  PROQA_HIP(hipEventCreate(&e1));
  int* sink = nullptr;
  PROQA_HIP(hipMalloc((void**)&sink, 4));
  const unsigned grid = (unsigned)device_cu_count() * 2;
  // (a 16x16x64 instruction is half the passes of a 32x32x32 one: twice the iterations for the same launch length)
  int iters = (int)(ms_target * 1e-3 * 2.4e9 / (4 * 2 * 32)) * (shape == 1 ? 2 : 1);
  if (iters < 64) iters = 64;
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    PROQA_HIP(hipEventRecord(e0, st));
    if (shape == 1)
      hipLaunchKernelGGL(mfma_loop_i8_16x16x64, dim3(grid), dim3(256), 0, st, iters, zero_operands, sink);
    else
      hipLaunchKernelGGL(mfma_loop_i8, dim3(grid), dim3(256), 0, st, iters, zero_operands, sink);
    PROQA_HIP(hipEventRecord(e1, st));
    PROQA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PROQA_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double ops = (double)grid * 4 /*waves*/ * (double)iters * 4 /*chains*/ * 2.0 * (shape == 1 ? 16.0 * 16 * 64 : 32.0 * 32 * 32);
  *tops = ops / (best * 1e-3) / 1e12;
  return PROQA_OK;
}

int proqa_microbench_mfma_i8(double ms_target, int zero_operands, void* stream, double* tops) {
  return proqa_microbench_mfma_i8_shape(ms_target, zero_operands, 0, stream, tops);
}

// microseconds of one cooperative launch of `grid` workgroups with n_syncs grid-wide barriers (best of 5 after a warm-up)
int proqa_microbench_grid_sync(int grid, int n_syncs, void* stream, double* us) {
  if (!us || grid <= 0 || grid > 4096 || n_syncs < 0) return fail(PROQA_EINVAL, "microbench_grid_sync: bad argument");
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  PROQA_HIP(hipEventCreate(&e0));
  PROQA_HIP(hipEventCreate(&e1));
  unsigned* buf = nullptr;
  PROQA_HIP(hipMalloc((void**)&buf, (size_t)grid * 64));
  PROQA_HIP(hipMemsetAsync(buf, 0, (size_t)grid * 64, st));
  float best = 1e30f;
  void* args[] = {&n_syncs, &buf};
  for (int r = 0; r < 6; ++r) {
    PROQA_HIP(hipEventRecord(e0, st));
    hipError_t e = hipLaunchCooperativeKernel((const void*)grid_sync_loop, dim3((unsigned)grid), dim3(256), args, 0, st);
    if (e != hipSuccess) {
      (void)hipFree(buf);
      return hip_fail(e, "hipLaunchCooperativeKernel", __FILE__, __LINE__);
    }
    PROQA_HIP(hipEventRecord(e1, st));
    PROQA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PROQA_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  (void)hipFree(buf);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *us = best * 1e3;
  return PROQA_OK;
}

// int8 MFMA rate (TOP/s, random operands) of a loop of 8 MFMAs + n_valu plain VALU instructions per wave, four waves per
// SIMD: n_valu in {0, 8, 16, 24, 32, 48, 64}
int proqa_microbench_mfma_i8_valu(double ms_target, int n_valu, void* stream, double* tops) {
  if (!tops || !(ms_target > 0)) return fail(PROQA_EINVAL, "microbench_mfma_i8_valu: bad argument");
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  PROQA_HIP(hipEventCreate(&e0));
  PROQA_HIP(hipEventCreate(&e1));
  int* sink = nullptr;
  PROQA_HIP(hipMalloc((void**)&sink, 4));
  const unsigned grid = (unsigned)device_cu_count() * 2;   // two 8-wave workgroups per CU: four waves per SIMD
  int iters = (int)(ms_target * 1e-3 * 2.4e9 / (8 * 4 * 32));
  if (iters < 64) iters = 64;
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    PROQA_HIP(hipEventRecord(e0, st));
    switch (n_valu) {
#define PROQA_CASE(N) case N: hipLaunchKernelGGL(mfma_i8_valu_loop<N>, dim3(grid), dim3(512), 0, st, iters, sink); break;
      PROQA_CASE(0) PROQA_CASE(8) PROQA_CASE(16) PROQA_CASE(24) PROQA_CASE(32) PROQA_CASE(48) PROQA_CASE(64)
#undef PROQA_CASE
      default:
        (void)hipFree(sink);
        return fail(PROQA_EINVAL, "microbench_mfma_i8_valu: n_valu=%d is not built (0, 8, 16, 24, 32, 48, 64)", n_valu);
    }
    PROQA_HIP(hipEventRecord(e1, st));
    PROQA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PROQA_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double ops = (double)grid * 8 /*waves*/ * (double)iters * 8 /*MFMAs*/ * 2.0 * 32 * 32 * 32;
  *tops = ops / (best * 1e-3) / 1e12;
  return PROQA_OK;
}

}  // extern "C"

// k-means on passage embeddings (proqa_kmeans_* in proqa_hip.h).
//
// Replaces faiss.Clustering.train + the final index.search(data, 1) of
// /root/reference/retrieval/group_paras.py:20-53 (IndexFlatL2, or IndexFlatIP when --spherical).
// The Lloyd loop itself (initialisation, sub-sampling, empty-cluster splitting) lives on the host
// (proqa_amd/group_paras.py); the two heavy steps are here:
//
//   assign   nearest centroid of every point.  Same MFMA structure as the top-k search with the
//            roles fixed the other way round: 512 POINTS per workgroup are resident as MFMA B
//            fragments (each lane owns one point), the CENTROIDS stream through LDS as the A
//            operand, and the lane keeps a running (best score, best centroid) — argmax instead
//            of a threshold test.  Centroids are fp32 in faiss; to keep fp32-grade dot products on
//            the fp16 matrix pipe every centroid row is split into hi + lo fp16 parts (x.c =
//            x.c_hi + x.c_lo, 16 k-steps) and, for L2, one extra k-step carries -|c|^2/2 split
//            into three fp16 terms against a constant 1 in the point operand:
//            argmin |x-c|^2 = argmax (x.c - |c|^2/2).
//   update   new centroid = mean of its points, accumulated in fp32 IN POINT ORDER (the order of
//            faiss' km_update_centroids), so equal assignments give bit-equal centroids: points
//            are stably sorted by assignment (hipCUB radix sort), one wave per centroid then adds
//            its rows sequentially.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <new>

#include "common.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = PROQA_EMBED_DIM;            // 128
// hi | lo | norm step | pad = 280 fp16 per centroid row: the 560-byte row stride (140 dwords, 12 mod
// 64 banks per row) makes the 16 rows of a ds_read_b128 lane group start on 16 distinct 4-bank
// slots, so the LDS image needs no swizzle
constexpr int kOpCols = 2 * kD + 16 + 8;
constexpr int kOpRowBytes = kOpCols * 2;       // 560 B = 35 pieces of 16 B
constexpr int kPieces = kOpRowBytes / 16;      // 35
constexpr int kKSteps = (2 * kD + 16) / 16;    // 17
constexpr int kStageRowsKm = 64;               // centroid rows per LDS stage
constexpr int kStageBytesKm = kStageRowsKm * kOpRowBytes;  // 35840
constexpr int kAssignWaves = 8;
constexpr int kAssignThreads = kAssignWaves * 64;
constexpr int kPointsPerBlock = kAssignWaves * 64;         // 2 blocks of 32 points per wave

// fp32 centroids -> MFMA operand rows [k_pad][280] fp16: c_hi | c_lo | (-|c|^2/2 as 3 fp16) | 0
__global__ void prep_centroids(const float* __restrict__ c, int k, int k_pad, int l2, _Float16* __restrict__ op) {
  const int row = blockIdx.x;
  const int t = threadIdx.x;  // 128 threads
  __shared__ float red[128];
  float v = 0.f;
  if (row < k) v = c[(long long)row * kD + t];
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  _Float16* dst = op + (long long)row * kOpCols;
  dst[t] = hi;
  dst[kD + t] = lo;
  // the norm uses the values the matrix pipe will actually see (hi + lo)
  const float seen = (float)hi + (float)lo;
  red[t] = seen * seen;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t < 24) {  // norm step (16) + row padding (8)
    float term = 0.f;
    if (l2 && row < k) {
      const float h = -0.5f * red[0];
      const _Float16 a = (_Float16)h;
      const _Float16 b = (_Float16)(h - (float)a);
      const _Float16 cc = (_Float16)(h - (float)a - (float)b);
      term = t == 0 ? (float)a : t == 1 ? (float)b : t == 2 ? (float)cc : 0.f;
    }
    dst[2 * kD + t] = (_Float16)term;
  }
}

template <bool L2>
__global__ __launch_bounds__(kAssignThreads) void kmeans_assign(const _Float16* __restrict__ x, long long n,
                                                                const char* __restrict__ op, int k, int k_pad,
                                                                int* __restrict__ out_idx,
                                                                float* __restrict__ out_dist) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kStageBytesKm];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const long long p0 = (long long)blockIdx.x * kPointsPerBlock + wave * 64;

  // resident point fragments (MFMA B operand) + |x|^2 of the lane's half rows
  f16x8 qf[2][8];
  float xx[2] = {0.f, 0.f};
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    long long p = p0 + blk * 32 + li;
    if (p >= n) p = n - 1;
    const _Float16* row = x + p * kD;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qf[blk][j] = *(const f16x8*)(row + (2 * j + half) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) xx[blk] += (float)qf[blk][j][e] * (float)qf[blk][j][e];
    }
  }
  // operand of the norm k-step: lanes of half 0 multiply the three norm terms by 1
  f16x8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    ones[0] = (_Float16)1.0f;
    ones[1] = (_Float16)1.0f;
    ones[2] = (_Float16)1.0f;
  }

  float best_s[2] = {-__builtin_inff(), -__builtin_inff()};
  int best_i[2] = {0, 0};

  const int nstages = k_pad / kStageRowsKm;
  auto issue_stage = [&](int s) {
    char* buf = lds + (s & 1) * kStageBytesKm;
    const char* src0 = op + (long long)s * kStageBytesKm;
    // 64 rows x 35 pieces = 35 wave-instructions of 1 KiB (a straight copy), round-robin over the waves
    for (int e = wave; e < kStageRowsKm * kPieces / 64; e += kAssignWaves) {
      const char* src = src0 + (e * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + e * 1024), 16, 0, 0);
    }
  };

  issue_stage(0);
  for (int s = 0; s < nstages; ++s) {
    dma_wait_barrier();  // stage s landed (explicit vmcnt 0), every wave left the other buffer
    if (s + 1 < nstages) issue_stage(s + 1);
    const char* buf = lds + (s & 1) * kStageBytesKm;
#pragma unroll
    for (int sub = 0; sub < kStageRowsKm / 32; ++sub) {
      const int row = sub * 32 + li;
      const char* rbase = buf + row * kOpRowBytes;
      f32x16 acc[2] = {{0}, {0}};
#pragma unroll
      for (int j = 0; j < kKSteps; ++j) {
        const f16x8 a = *(const f16x8*)(rbase + (2 * j + half) * 16);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const f16x8 b = j < 8 ? qf[blk][j] : j < 16 ? qf[blk][j - 8] : ones;
          if (j < 16 || L2) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[blk], 0, 0, 0);
        }
      }
      const int crow0 = s * kStageRowsKm + sub * 32 + 4 * half;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        float m = acc[blk][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = __builtin_fmaxf(m, acc[blk][r]);
        if (__any(m > best_s[blk])) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {  // ascending centroid index: strict > keeps the lowest on ties
            const int c = crow0 + (r & 3) + 8 * (r >> 2);
            if (acc[blk][r] > best_s[blk] && c < k) {
              best_s[blk] = acc[blk][r];
              best_i[blk] = c;
            }
          }
        }
      }
    }
  }

#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    // combine the two accumulator halves of the point (they saw interleaved centroid rows)
    const float os = __shfl_xor(best_s[blk], 32, 64);
    const int oi = __shfl_xor(best_i[blk], 32, 64);
    if (os > best_s[blk] || (os == best_s[blk] && oi < best_i[blk])) {
      best_s[blk] = os;
      best_i[blk] = oi;
    }
    const float xx_all = xx[blk] + __shfl_xor(xx[blk], 32, 64);
    const long long p = p0 + blk * 32 + li;
    if (half == 0 && p < n) {
      out_idx[p] = best_i[blk];
      // faiss IndexFlatL2 reports |x|^2 + |c|^2 - 2 x.c clipped at 0; IndexFlatIP the inner product
      out_dist[p] = L2 ? __builtin_fmaxf(xx_all - 2.0f * best_s[blk], 0.f) : best_s[blk];
    }
  }
}

__global__ void histogram_assign(const int* __restrict__ assign, long long n, unsigned* __restrict__ counts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&counts[assign[i]], 1u);
}

__global__ void iota_u32(unsigned* v, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (unsigned)i;
}

// exclusive scan of k counters by one workgroup (k is at most a few 10^5)
__global__ __launch_bounds__(1024) void exclusive_scan_counts(const unsigned* __restrict__ counts, int k,
                                                              unsigned* __restrict__ begin) {
  __shared__ unsigned part[1024];
  const int t = threadIdx.x;
  const int per = (k + 1023) / 1024;
  const int lo = t * per, hi = min(k, lo + per);
  unsigned s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned v = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned run = t ? part[t - 1] : 0u;
  for (int i = lo; i < hi; ++i) {
    begin[i] = run;
    run += counts[i];
  }
}

// one wave per centroid: c = (sum of its rows, added in point order, fp32) / count
__global__ __launch_bounds__(256) void segmented_mean(const _Float16* __restrict__ x, const unsigned* __restrict__ order,
                                                      const unsigned* __restrict__ begin,
                                                      const unsigned* __restrict__ counts, int k,
                                                      float* __restrict__ centroids) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= k) return;
  const unsigned n = counts[c];
  if (n == 0) return;  // empty cluster: the host re-seeds it (faiss splits a large cluster)
  const unsigned b = begin[c];
  float s0 = 0.f, s1 = 0.f;
  // the additions are sequential (point order), the loads are not: 32 rows in flight per wave, so that a cluster that
  // attracted a hundred times its share of the points (first iterations; hubs of real embeddings) is bound by memory
  // throughput, not by one load latency per point
  constexpr unsigned kInFlight = 32;
  for (unsigned i0 = 0; i0 < n; i0 += 64) {
    const unsigned mine = i0 + lane < n ? order[b + i0 + lane] : 0u;
    const unsigned m = min(64u, n - i0);
    for (unsigned j0 = 0; j0 < m; j0 += kInFlight) {
      f16x2 v[kInFlight];
#pragma unroll
      for (unsigned u = 0; u < kInFlight; ++u) {
        const unsigned p = __shfl(mine, (int)min(j0 + u, m - 1), 64);   // past the end: the last row again, not added
        v[u] = *(const f16x2*)(x + (long long)p * kD + 2 * lane);
      }
#pragma unroll
      for (unsigned u = 0; u < kInFlight; ++u) {
        if (j0 + u < m) {
          s0 += (float)v[u][0];
          s1 += (float)v[u][1];
        }
      }
    }
  }
  const float inv = (float)n;
  centroids[(long long)c * kD + 2 * lane] = s0 / inv;
  centroids[(long long)c * kD + 2 * lane + 1] = s1 / inv;
}

}  // namespace
}  // namespace proqa

struct proqa_kmeans {
  int device = 0;
  int64_t n_max = 0;
  int k = 0, k_pad = 0;
  _Float16* op = nullptr;       // centroid operand rows
  unsigned* counts = nullptr;   // [k]
  unsigned* begin = nullptr;    // [k]
  unsigned* keys_in = nullptr;  // [n_max] (assignment as unsigned)
  unsigned* keys_out = nullptr;
  unsigned* vals_in = nullptr;
  unsigned* vals_out = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
};

using namespace proqa;

extern "C" {

int proqa_kmeans_free(proqa_kmeans* h) {
  if (!h) return PROQA_OK;
  void* ptrs[] = {h->op, h->counts, h->begin, h->keys_in, h->keys_out, h->vals_in, h->vals_out, h->sort_tmp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete h;
  return PROQA_OK;
}

int proqa_kmeans_create(int d, int64_t n_max, int k, proqa_kmeans** out) {
  if (!out) return fail(PROQA_EINVAL, "kmeans_create: out is NULL");
  *out = nullptr;
  if (d != kD) return fail(PROQA_EINVAL, "kmeans_create: d=%d, only d=128 is supported", d);
  if (n_max <= 0 || n_max >= (1ll << 31) || k <= 0 || k > (1 << 20))
    return fail(PROQA_EINVAL, "kmeans_create: n_max=%lld k=%d out of range", (long long)n_max, k);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PROQA_ENOGPU, "kmeans_create: no HIP device");
  proqa_kmeans* h = new (std::nothrow) proqa_kmeans();
  if (!h) return fail(PROQA_ENOMEM, "kmeans_create: out of host memory");
  PROQA_HIP(hipGetDevice(&h->device));
  h->n_max = n_max;
  h->k = k;
  h->k_pad = round_up<int>(k, kStageRowsKm);
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes);
  };
  alloc((void**)&h->op, (size_t)h->k_pad * kOpRowBytes);
  alloc((void**)&h->counts, (size_t)k * sizeof(unsigned));
  alloc((void**)&h->begin, (size_t)k * sizeof(unsigned));
  alloc((void**)&h->keys_in, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->keys_out, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->vals_in, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->vals_out, (size_t)n_max * sizeof(unsigned));
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, h->sort_tmp_bytes, h->keys_in, h->keys_out, h->vals_in, h->vals_out,
                                           (int)n_max, 0, 32, nullptr);
  alloc(&h->sort_tmp, h->sort_tmp_bytes);
  if (e != hipSuccess) {
    proqa_kmeans_free(h);
    return fail(PROQA_ENOMEM, "kmeans_create: device allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return PROQA_OK;
}

int proqa_kmeans_assign_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const float* centroids_dev,
                               int metric_l2, int32_t* assign_dev, float* dist_dev, void* stream) {
  if (!h || !x_f16_dev || !centroids_dev || !assign_dev || !dist_dev) return fail(PROQA_EINVAL, "kmeans_assign: NULL argument");
  if (n < 0 || n >= (1ll << 31)) return fail(PROQA_EINVAL, "kmeans_assign: n=%lld", (long long)n);
  if (n == 0) return PROQA_OK;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(prep_centroids, dim3(h->k_pad), dim3(128), 0, st, centroids_dev, h->k, h->k_pad, metric_l2 ? 1 : 0,
                     h->op);
  PROQA_LAUNCH_CHECK();
  const unsigned grid = (unsigned)ceil_div<int64_t>(n, kPointsPerBlock);
  if (metric_l2)
    hipLaunchKernelGGL((kmeans_assign<true>), dim3(grid), dim3(kAssignThreads), 0, st, (const _Float16*)x_f16_dev,
                       (long long)n, (const char*)h->op, h->k, h->k_pad, assign_dev, dist_dev);
  else
    hipLaunchKernelGGL((kmeans_assign<false>), dim3(grid), dim3(kAssignThreads), 0, st, (const _Float16*)x_f16_dev,
                       (long long)n, (const char*)h->op, h->k, h->k_pad, assign_dev, dist_dev);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

int proqa_kmeans_update_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const int32_t* assign_dev,
                               float* centroids_dev, uint32_t* counts_dev, void* stream) {
  if (!h || !x_f16_dev || !assign_dev || !centroids_dev || !counts_dev)
    return fail(PROQA_EINVAL, "kmeans_update: NULL argument");
  if (n <= 0 || n > h->n_max) return fail(PROQA_EINVAL, "kmeans_update: n=%lld exceeds n_max=%lld", (long long)n, (long long)h->n_max);
  hipStream_t st = as_stream(stream);
  const unsigned blocks = (unsigned)ceil_div<int64_t>(n, 256);
  PROQA_HIP(hipMemsetAsync(h->counts, 0, (size_t)h->k * sizeof(unsigned), st));
  hipLaunchKernelGGL(histogram_assign, dim3(blocks), dim3(256), 0, st, assign_dev, (long long)n, h->counts);
  PROQA_LAUNCH_CHECK();
  hipLaunchKernelGGL(exclusive_scan_counts, dim3(1), dim3(1024), 0, st, h->counts, h->k, h->begin);
  PROQA_LAUNCH_CHECK();
  hipLaunchKernelGGL(iota_u32, dim3(blocks), dim3(256), 0, st, h->vals_in, (long long)n);
  PROQA_LAUNCH_CHECK();
  int bits = 1;
  while ((1 << bits) < h->k) ++bits;
  size_t tmp = h->sort_tmp_bytes;
  PROQA_HIP(hipcub::DeviceRadixSort::SortPairs(h->sort_tmp, tmp, (const unsigned*)assign_dev, h->keys_out, h->vals_in,
                                               h->vals_out, (int)n, 0, bits, st));
  hipLaunchKernelGGL(segmented_mean, dim3((unsigned)ceil_div<int>(h->k, 4)), dim3(256), 0, st, (const _Float16*)x_f16_dev,
                     h->vals_out, h->begin, h->counts, h->k, centroids_dev);
  PROQA_LAUNCH_CHECK();
  PROQA_HIP(hipMemcpyAsync(counts_dev, h->counts, (size_t)h->k * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
  return PROQA_OK;
}

}  // extern "C"

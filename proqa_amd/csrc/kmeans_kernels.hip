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
//            Two passes: the hi parts alone nominate (half the k-steps; the lo parts move a score by at most
//            |x| max|c_lo|, so a lead larger than twice that decides), the full-precision kernel runs over the few
//            per cent of the points the nomination leaves undecided -- see kmeans_assign's MODEs.
//   update   new centroid = mean of its points, accumulated in fp32 IN POINT ORDER (the order of
//            faiss' km_update_centroids), so equal assignments give bit-equal centroids: points
//            are stably sorted by assignment (hipCUB radix sort), one wave per centroid then adds
//            its rows sequentially.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstring>
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
constexpr int kKSteps = (2 * kD + 16) / 16;    // 17
constexpr int kStageRowsKm = 64;               // centroid rows per LDS stage
// (64 rows x 560 B = 35 KiB per LDS stage of the full operand)
// the nominating pass of the assignment streams hi | norm step | pad only: 152 fp16 = 304 B per row (76 dwords: the same
// 12-mod-64 bank stride as the full row), 19 KiB per stage -- half the operand traffic for half the k-steps
constexpr int kHiCols = kD + 16 + 8;
constexpr int kHiRowBytes = kHiCols * 2;                   // 304
constexpr int kAssignWaves = 8;
constexpr int kAssignThreads = kAssignWaves * 64;
constexpr int kPointsPerBlock = kAssignWaves * 64;         // 2 blocks of 32 points per wave

// fp32 centroids -> MFMA operand rows [k_pad][280] fp16: c_hi | c_lo | (-|c|^2/2 as 3 fp16) | 0
// stats (optional): {max_c |c_lo|^2, max_c |c_hi + c_lo|^2} as float bits (non-negative: unsigned order = float order),
// zeroed by the caller -- what the nominating pass of the assignment bounds its error with
__global__ void prep_centroids(const float* __restrict__ c, int k, int k_pad, int l2, _Float16* __restrict__ op,
                               unsigned* __restrict__ stats, _Float16* __restrict__ op_hi) {
  const int row = blockIdx.x;
  const int t = threadIdx.x;  // 128 threads
  __shared__ float red[128];
  __shared__ float red_lo[128];
  float v = 0.f;
  if (row < k) v = c[(long long)row * kD + t];
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  _Float16* dst = op + (long long)row * kOpCols;
  dst[t] = hi;
  dst[kD + t] = lo;
  _Float16* dst_hi = op_hi ? op_hi + (long long)row * kHiCols : nullptr;
  if (dst_hi) dst_hi[t] = hi;
  // the norm uses the values the matrix pipe will actually see (hi + lo)
  const float seen = (float)hi + (float)lo;
  red[t] = seen * seen;
  red_lo[t] = (float)lo * (float)lo;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (t < s) {
      red[t] += red[t + s];
      red_lo[t] += red_lo[t + s];
    }
    __syncthreads();
  }
  if (stats && t == 0 && row < k) {
    atomicMax(stats, __float_as_uint(red_lo[0]));
    atomicMax(stats + 1, __float_as_uint(red[0]));
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
    if (dst_hi) dst_hi[kD + t] = (_Float16)term;
  }
}

// ds_read_b128 and its counted wait as inline assembly: hipcc sinks every fragment load of a unit to its use, into ONE
// register, and waits lgkmcnt(0) in front of each pair of MFMAs -- an LDS round trip per k-step, which two waves per SIMD do
// not cover.  A read issued this way is invisible to the compiler's wait insertion; lds_landed<N>() names the fragment it
// retires (reads return in order: N younger ones may still be in flight), so its use is ordered after the wait.
template <int IMM>
__device__ __forceinline__ f16x8 lds_read128(unsigned addr) {
  f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lds_landed(f16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}
// the k-steps of one unit: every fragment requested up front, each MFMA pair behind the wait for its own fragment
template <int J, int STEPS, bool L2>
struct UnitSteps {
  static __device__ __forceinline__ void read(f16x8 (&fa)[STEPS], unsigned addr) {
    fa[J] = lds_read128<J * 32>(addr);
    UnitSteps<J + 1, STEPS, L2>::read(fa, addr);
  }
  static __device__ __forceinline__ void run(f16x8 (&fa)[STEPS], const f16x8 (&qf)[2][8], const f16x8& ones, f32x16 (&acc)[2]) {
    constexpr bool norm_step = J == STEPS - 1;
    lds_landed<(STEPS - 1 - J < 15 ? STEPS - 1 - J : 15)>(fa[J]);   // (the counter holds 15: a stricter wait for the first of 17)
    if (!norm_step || L2) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
        acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[J], norm_step ? ones : qf[blk][J & 7], acc[blk], 0, 0, 0);
    }
    UnitSteps<J + 1, STEPS, L2>::run(fa, qf, ones, acc);
  }
};
template <int STEPS, bool L2>
struct UnitSteps<STEPS, STEPS, L2> {
  static __device__ __forceinline__ void read(f16x8 (&)[STEPS], unsigned) {}
  static __device__ __forceinline__ void run(f16x8 (&)[STEPS], const f16x8 (&)[2][8], const f16x8&, f32x16 (&)[2]) {}
};

// MODE_FULL: the assignment at full precision (hi + lo + norm: 17 k-steps) of points [0, n).
// MODE_NOMINATE: hi + norm only (9 k-steps).  A centroid's score is then off by x.c_lo, at most |x| max|c_lo| in size, so
//   the best centroid of this pass is THE best one whenever it leads the runner-up by more than twice that (+ a generous
//   allowance for fp32 summation); the pass keeps best and runner-up, reports the leader with its score corrected by the
//   exact x.c_lo of that one centroid, and appends the points it cannot decide (a few per cent; ties always) to `sel`.
// MODE_SELECTED: MODE_FULL over the points listed in sel[0, *sel_count), persistent workgroups.
enum { MODE_FULL = 0, MODE_NOMINATE = 1, MODE_SELECTED = 2 };

template <bool L2, int MODE>
__global__ __launch_bounds__(kAssignThreads) void kmeans_assign(const _Float16* __restrict__ x, long long n,
                                                                const char* __restrict__ op, int k, int k_pad,
                                                                int* __restrict__ out_idx,
                                                                float* __restrict__ out_dist, unsigned* __restrict__ sel,
                                                                unsigned* __restrict__ sel_count,
                                                                const unsigned* __restrict__ stats,
                                                                const char* __restrict__ op_hi,
                                                                const int* __restrict__ hint,
                                                                const unsigned* __restrict__ order) {
  constexpr int kRowB = MODE == MODE_NOMINATE ? kHiRowBytes : kOpRowBytes;
  constexpr int kStageB = kStageRowsKm * kRowB;
  constexpr int kSteps = MODE == MODE_NOMINATE ? 9 : kKSteps;   // hi (8) [+ lo (8)] + norm (1)
  // Ring of four stages, the DMA of three in flight: a stage of the nominating pass is ~1 us of MFMA work, less than one
  // trip to the L2 / Infinity Cache the operand lives in.  Every wave issues the SAME number of 1 KiB pieces per stage (a
  // wave whose share is one short repeats its last piece), so one counted vmcnt per stage says "stage s has landed".
  constexpr int kRing = 4;
  constexpr int kStagePieces = kStageB / 1024;                                      // 35 (19)
  constexpr int kPiecesPerWave = (kStagePieces + kAssignWaves - 1) / kAssignWaves;  // 5 (3)
  __shared__ __attribute__((aligned(16))) char lds[kRing * kStageB];
  const char* stream_op = MODE == MODE_NOMINATE ? op_hi : op;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  if (MODE == MODE_SELECTED) n = (long long)*sel_count;
  // operand of the norm k-step: lanes of half 0 multiply the three norm terms by 1
  f16x8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    ones[0] = (_Float16)1.0f;
    ones[1] = (_Float16)1.0f;
    ones[2] = (_Float16)1.0f;
  }
  const int nstages = k_pad / kStageRowsKm;
  auto issue_stage = [&](int s) {
    char* buf = lds + (s & (kRing - 1)) * kStageB;
    const char* src0 = stream_op + (long long)s * kStageB;
    // 64 rows x 35 (19) pieces of 1 KiB (a straight copy), round-robin over the waves
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) {
      int e = wave + i * kAssignWaves;
      if (e >= kStagePieces) e -= kAssignWaves;   // this wave's share is one short: its last piece once more
      const char* src = src0 + (e * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + e * 1024), 16, 0, 0);
    }
  };
  // stage s has landed (the pieces of at most `younger` later stages stay in flight) and every wave has left stage s - 1
  auto publish = [&](bool full_depth) {
    if (full_depth)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRing - 2) * kPiecesPerWave) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  for (long long tile = blockIdx.x; tile * kPointsPerBlock < n; tile += gridDim.x) {
    const long long p0 = tile * kPointsPerBlock + wave * 64;
    // resident point fragments (MFMA B operand) + |x|^2 of the lane's half rows
    f16x8 qf[2][8];
    float xx[2] = {0.f, 0.f};
    long long prow[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      long long p = p0 + blk * 32 + li;
      if (p >= n) p = n - 1;
      // MODE_NOMINATE with `order` (the points sorted by their previous centroid, a by-product of the last update): the
      // 64 points of a wave then share one or two leaders, so the units in which SOME lane has bookkeeping to do are a
      // handful per wave instead of one or two per lane -- any permutation of the points is a valid processing order
      prow[blk] = MODE == MODE_SELECTED ? (long long)sel[p] : (MODE == MODE_NOMINATE && order ? (long long)order[p] : p);
      const _Float16* row = x + prow[blk] * kD;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        qf[blk][j] = *(const f16x8*)(row + (2 * j + half) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xx[blk] += (float)qf[blk][j][e] * (float)qf[blk][j][e];
      }
    }

    float best_s[2] = {-__builtin_inff(), -__builtin_inff()};
    float second_s[2] = {-__builtin_inff(), -__builtin_inff()};   // MODE_NOMINATE
    // MODE_NOMINATE: a score is off by x.c_lo, at most |x| max|c_lo|; a lead of twice that (+ a generous allowance for
    // the fp32 summation) decides
    float margin[2] = {0.f, 0.f};
    if (MODE == MODE_NOMINATE) {
      const float lo_max = __builtin_sqrtf(__uint_as_float(stats[0])), c_max = __builtin_sqrtf(__uint_as_float(stats[1]));
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const float xn = __builtin_sqrtf(xx[blk] + __shfl_xor(xx[blk], 32, 64));
        margin[blk] = 2.0f * (xn * lo_max + 4e-5f * (xn * c_max + 0.5f * c_max * c_max + 1.0f));
      }
    }
    // MODE_NOMINATE with a hint (the point's centroid of the previous Lloyd iteration; any value allowed): the hinted
    // centroid's score, computed here on the vector pipe, less twice the margin is below the final leader's and further than
    // the margin from it -- the pass starts with that as its (virtual) leader, so nothing below it costs any bookkeeping
    int best_i[2] = {0, 0};
    if (MODE == MODE_NOMINATE && hint) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int hc = hint[prow[blk]];
        const bool ok = hc >= 0 && hc < k;
        float dot = 0.f;
        if (ok) {
          const _Float16* crow = (const _Float16*)(op_hi + (long long)hc * kHiRowBytes);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f16x8 ch = *(const f16x8*)(crow + (2 * j + half) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += (float)qf[blk][j][e] * (float)ch[e];
          }
          if (L2 && half == 0) dot += (float)crow[kD] + (float)crow[kD + 1] + (float)crow[kD + 2];   // -|c|^2 / 2
        }
        dot += __shfl_xor(dot, 32, 64);
        // (the matrix pipe sums the same products in another order: the margin holds twice that allowance)
        if (ok) {
          best_s[blk] = dot - 2.0f * margin[blk];
          best_i[blk] = hc;
        }
      }
    }

    if (MODE == MODE_SELECTED && tile != (long long)blockIdx.x) __syncthreads();   // every wave has left the previous tile's buffers
#pragma unroll
    for (int s = 0; s < kRing - 1; ++s)
      if (s < nstages) issue_stage(s);
    for (int s = 0; s < nstages; ++s) {
      publish(s + kRing - 2 < nstages);   // (near the end fewer stages are in flight: wait for all)
      if (s + kRing - 1 < nstages) issue_stage(s + kRing - 1);
#pragma unroll
      for (int sub = 0; sub < kStageRowsKm / 32; ++sub) {
        const int row = sub * 32 + li;
        f32x16 acc[2] = {{0}, {0}};
        f16x8 fa[kSteps];
        const unsigned frag_addr = lds0 + (unsigned)((s & (kRing - 1)) * kStageB + row * kRowB + half * 16);
        UnitSteps<0, kSteps, L2>::read(fa, frag_addr);
        UnitSteps<0, kSteps, L2>::run(fa, qf, ones, acc);
        const int crow0 = s * kStageRowsKm + sub * 32 + 4 * half;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          if (MODE == MODE_NOMINATE && (s + 1) * kStageRowsKm > k) {   // wave-uniform: operand padding rows (>= k), last stage only
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (crow0 + (r & 3) + 8 * (r >> 2) >= k) acc[blk][r] = -__builtin_inff();
          }
          float m = acc[blk][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) m = __builtin_fmaxf(m, acc[blk][r]);
          if (MODE == MODE_NOMINATE) {
            // A runner-up in ANOTHER unit than the leader's shows in that unit's maximum: two operations per unit, no
            // branch.  Only a unit that takes the lead needs more -- its second-best score (a tournament over the 16: 37
            // operations) and the index of its best (the lowest centroid with that score) -- and with the leader starting
            // at the hinted centroid's score that happens once or twice per point, in the same units for the whole wave
            // when the points arrive sorted by their previous centroid.
            second_s[blk] = __builtin_fmaxf(second_s[blk], __builtin_fminf(best_s[blk], m));
            if (__any(m > best_s[blk])) {
              float v[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] = acc[blk][r];
              float hi8[8], lo8[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                hi8[i] = __builtin_fmaxf(v[2 * i], v[2 * i + 1]);
                lo8[i] = __builtin_fminf(v[2 * i], v[2 * i + 1]);
              }
#pragma unroll
              for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                for (int i = 0; i < w; ++i) {
                  const float a1 = hi8[i], b1 = hi8[i + w];
                  lo8[i] = __builtin_fmaxf(__builtin_fmaxf(lo8[i], lo8[i + w]), __builtin_fminf(a1, b1));
                  hi8[i] = __builtin_fmaxf(a1, b1);
                }
              const float m2 = lo8[0];   // (hi8[0] == m)
              int first = 0;
#pragma unroll
              for (int r = 15; r >= 0; --r) first = v[r] == m ? crow0 + (r & 3) + 8 * (r >> 2) : first;
              const bool lead = m > best_s[blk];   // strict: an equal score in a later unit stays the runner-up
              second_s[blk] = lead ? __builtin_fmaxf(second_s[blk], m2) : second_s[blk];
              best_i[blk] = lead ? first : best_i[blk];
              best_s[blk] = __builtin_fmaxf(best_s[blk], m);
            }
          } else if (__any(m > best_s[blk])) {
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
      const float o2 = __shfl_xor(second_s[blk], 32, 64);
      float runner = __builtin_fmaxf(__builtin_fminf(os, best_s[blk]), __builtin_fmaxf(o2, second_s[blk]));
      if (os > best_s[blk] || (os == best_s[blk] && oi < best_i[blk])) {
        best_s[blk] = os;
        best_i[blk] = oi;
      }
      const float xx_all = xx[blk] + __shfl_xor(xx[blk], 32, 64);
      const long long p = p0 + blk * 32 + li;
      float score = best_s[blk];
      if (MODE == MODE_NOMINATE) {
        // the leader's score, corrected by the lo part of that one centroid (each half adds its 64 dimensions)
        const _Float16* lo = (const _Float16*)(op + (long long)best_i[blk] * kOpRowBytes) + kD;
        float corr = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f16x8 cl = *(const f16x8*)(lo + (2 * j + half) * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) corr += (float)qf[blk][j][e] * (float)cl[e];
        }
        corr += __shfl_xor(corr, 32, 64);
        score += corr;
        // undecided: the runner-up is within the margin of the leader
        if (half == 0 && p < n && !(best_s[blk] - runner > margin[blk])) sel[atomicAdd(sel_count, 1u)] = (unsigned)prow[blk];
      }
      if (half == 0 && p < n) {
        out_idx[prow[blk]] = best_i[blk];
        // faiss IndexFlatL2 reports |x|^2 + |c|^2 - 2 x.c clipped at 0; IndexFlatIP the inner product
        out_dist[prow[blk]] = L2 ? __builtin_fmaxf(xx_all - 2.0f * score, 0.f) : score;
      }
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
  _Float16* op_hi = nullptr;    // the same without the lo parts (nominating pass)
  unsigned* counts = nullptr;   // [k]
  unsigned* begin = nullptr;    // [k]
  unsigned* keys_in = nullptr;  // [n_max] (assignment as unsigned)
  unsigned* keys_out = nullptr;
  unsigned* vals_in = nullptr;
  unsigned* vals_out = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  unsigned* words = nullptr;    // {max |c_lo|^2, max |c|^2 (float bits), undecided points} of the two-pass assignment
  int64_t order_n = 0;          // vals_out holds the ids of this many points sorted by assignment (the last update's)
};

using namespace proqa;

extern "C" {

int proqa_kmeans_free(proqa_kmeans* h) {
  if (!h) return PROQA_OK;
  void* ptrs[] = {h->op, h->counts, h->begin, h->keys_in, h->keys_out, h->vals_in, h->vals_out, h->sort_tmp, h->words, h->op_hi};
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
  alloc((void**)&h->op_hi, (size_t)h->k_pad * kHiRowBytes);
  alloc((void**)&h->counts, (size_t)k * sizeof(unsigned));
  alloc((void**)&h->begin, (size_t)k * sizeof(unsigned));
  alloc((void**)&h->keys_in, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->keys_out, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->vals_in, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->vals_out, (size_t)n_max * sizeof(unsigned));
  alloc((void**)&h->words, 4 * sizeof(unsigned));
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
  return proqa_kmeans_assign_hinted_device(h, x_f16_dev, n, centroids_dev, metric_l2, nullptr, assign_dev, dist_dev, stream);
}

int proqa_kmeans_assign_hinted_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const float* centroids_dev,
                                      int metric_l2, const int32_t* hint_dev, int32_t* assign_dev, float* dist_dev,
                                      void* stream) {
  if (!h || !x_f16_dev || !centroids_dev || !assign_dev || !dist_dev) return fail(PROQA_EINVAL, "kmeans_assign: NULL argument");
  if (n < 0 || n >= (1ll << 31)) return fail(PROQA_EINVAL, "kmeans_assign: n=%lld", (long long)n);
  if (n == 0) return PROQA_OK;
  hipStream_t st = as_stream(stream);
  // Two passes (the default wherever the handle's index buffer holds n points): hi-only nomination of every point, then the
  // full-precision kernel over the few it could not decide.  PROQA_KMEANS_TWO_PASS=0: the full-precision kernel over all.
  static const bool kTwoPass = !(getenv("PROQA_KMEANS_TWO_PASS") && atoi(getenv("PROQA_KMEANS_TWO_PASS")) == 0);
  const bool two_pass = kTwoPass && n <= h->n_max;
  // a hinted call right behind an update of the same n points (a Lloyd loop): walk the points in that update's sorted order
  static const bool kSorted = !(getenv("PROQA_KMEANS_SORTED") && atoi(getenv("PROQA_KMEANS_SORTED")) == 0);
  const unsigned* order = two_pass && hint_dev && kSorted && h->order_n == n ? h->vals_out : nullptr;
  PROQA_HIP(hipMemsetAsync(h->words, 0, 4 * sizeof(unsigned), st));
  hipLaunchKernelGGL(prep_centroids, dim3(h->k_pad), dim3(128), 0, st, centroids_dev, h->k, h->k_pad, metric_l2 ? 1 : 0,
                     h->op, h->words, two_pass ? h->op_hi : nullptr);
  PROQA_LAUNCH_CHECK();
  const unsigned grid = (unsigned)ceil_div<int64_t>(n, kPointsPerBlock);
  const _Float16* x = (const _Float16*)x_f16_dev;
#define PROQA_KM_LAUNCH(MODE, GRID)                                                                                          \
  do {                                                                                                                       \
    if (metric_l2)                                                                                                           \
      hipLaunchKernelGGL((kmeans_assign<true, MODE>), dim3(GRID), dim3(kAssignThreads), 0, st, x, (long long)n,              \
                         (const char*)h->op, h->k, h->k_pad, assign_dev, dist_dev, h->vals_in, h->words + 2, h->words,      \
                         (const char*)h->op_hi, hint_dev, order);                                                            \
    else                                                                                                                     \
      hipLaunchKernelGGL((kmeans_assign<false, MODE>), dim3(GRID), dim3(kAssignThreads), 0, st, x, (long long)n,             \
                         (const char*)h->op, h->k, h->k_pad, assign_dev, dist_dev, h->vals_in, h->words + 2, h->words,      \
                         (const char*)h->op_hi, hint_dev, order);                                                            \
    PROQA_LAUNCH_CHECK();                                                                                                    \
  } while (0)
  if (!two_pass) {
    PROQA_KM_LAUNCH(MODE_FULL, grid);
  } else {
    PROQA_KM_LAUNCH(MODE_NOMINATE, grid);
    static const bool kDebug = getenv("PROQA_DEBUG_KMEANS") != nullptr;   // developer: undecided points per call (adds a sync)
    if (kDebug) {
      unsigned w[4] = {};
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(w, h->words, sizeof w, hipMemcpyDeviceToHost);
      float lo2, c2;
      memcpy(&lo2, &w[0], 4);
      memcpy(&c2, &w[1], 4);
      fprintf(stderr, "[kmeans] nominating pass: %u of %lld points undecided (max |c_lo| %.3g, max |c| %.3g)\n", w[2], (long long)n,
              std::sqrt(lo2), std::sqrt(c2));
    }
    // (the count of undecided points stays on the device: a persistent grid walks over however many there are)
    PROQA_KM_LAUNCH(MODE_SELECTED, std::min<unsigned>(grid, 4u * (unsigned)device_cu_count()));
  }
#undef PROQA_KM_LAUNCH
  return PROQA_OK;
}

int proqa_kmeans_update_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const int32_t* assign_dev,
                               float* centroids_dev, uint32_t* counts_dev, void* stream) {
  if (!h || !x_f16_dev || !assign_dev || !centroids_dev || !counts_dev)
    return fail(PROQA_EINVAL, "kmeans_update: NULL argument");
  if (n <= 0 || n > h->n_max) return fail(PROQA_EINVAL, "kmeans_update: n=%lld exceeds n_max=%lld", (long long)n, (long long)h->n_max);
  hipStream_t st = as_stream(stream);
  const unsigned blocks = (unsigned)ceil_div<int64_t>(n, 256);
  h->order_n = 0;   // (set again once the sorted ids of this call are on the stream)
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
  h->order_n = n;
  return PROQA_OK;
}

}  // extern "C"

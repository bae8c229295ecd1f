// Exhaustive inner-product top-k over an fp16 [N,128] corpus resident in HBM.
//
// Replaces faiss.IndexFlatIP.search at /root/reference/retrieval/eval_retrieval.py:102-104.
//
// Structure (gfx950 / CDNA4, wave64):
//   mips_filter_f16   Q.P^T on MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) with the
//                     score matrix never leaving registers.  A workgroup of 8 waves keeps
//                     8*QW*32 queries as MFMA B-fragments in VGPRs for its whole lifetime and
//                     streams a contiguous chunk of corpus rows through a 2x32 KiB LDS ring
//                     filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction,
//                     XOR-swizzled on the SOURCE address so ds_read_b128 is conflict-free).
//                     Each lane owns one query column of the 32x32 accumulator, so the top-k
//                     test is lane-local: max of 16 scores vs. the query's running threshold,
//                     one wave-wide branch; survivors are appended to a per-query candidate
//                     list in HBM.
//   topk_merge        one workgroup per query: bitonic-sorts running top-k + candidates by
//                     (score desc, row asc) in LDS, keeps the best k, publishes the k-th
//                     score as the next round's threshold.
//   The host (mips_index.cpp) runs rounds over geometrically growing corpus slabs so the
//   threshold tightens quickly and later slabs produce only a few candidates per query.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mips_kernels.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = kDim * 2;                     // 256 B per fp16 corpus row
constexpr int kStageBytes = kStageRows * kRowBytes;     // 32 KiB
constexpr int kSubRows = 32;                            // one MFMA M-tile
constexpr int kSubBytes = kSubRows * kRowBytes;         // 8 KiB

__device__ __forceinline__ unsigned ord_from_float(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;  // -0.0 ties with +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_from_ord(unsigned o) {
  unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
// descending key order == (score descending, row ascending)
__device__ __forceinline__ unsigned long long pack_key(float score, unsigned row) {
  return ((unsigned long long)ord_from_float(score) << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}

// ---------------------------------------------------------------------------------------
// filter kernel
// ---------------------------------------------------------------------------------------
// Tiling: a workgroup = 8 waves; wave w keeps query blocks (32 queries each) w*QW .. w*QW+QW-1 of
// its query tile as MFMA B fragments in VGPRs and every wave reads the same corpus sub-tile
// (32 rows, MFMA A operand) from LDS.  The loop is arranged so that the MFMA pipe always has
// queued work from a single wave:
//   * unit = one 32-row sub-tile x QW query blocks (QW independent MFMA chains, interleaved);
//   * the A fragments of unit g+1 are read from LDS while unit g's chains run;
//   * the lane-local max-tree + threshold test of unit g-1 sits in the same basic block as unit
//     g's MFMAs (ONE wave-wide branch per unit, re-tested per tile only on the rare path);
//   * three LDS stage buffers and ONE barrier per stage placed mid-stage: crossing a stage
//     boundary needs no barrier, so the fragment prefetch runs straight across it.
//     At the barrier of stage s (before its unit 2): every wave is past stage s-1, whose buffer
//     (s+2)%3 is therefore free for DMA(s+2); and every wave has drained (vmcnt 0) its pieces of
//     DMA(s+1), issued one full stage earlier, so buffer (s+1)%3 is readable from unit 3 on.
// Candidate records.  A lane whose 16-score column beats its query's threshold appends ONE 80-byte
// record {query, first row, rows left in the chunk, 16 scores} to its wave's private log in HBM:
// the slot is the wave's running count (an SGPR) plus the lane's rank among the hit lanes, so the
// MFMA loop contains no atomics and never waits on memory.  When the wave has finished its chunk it
// drains its own log (drain_wave_log): one lane per record picks the individual scores that pass
// and appends them to the per-query candidate lists in HBM.
__device__ __forceinline__ void append_records(const f32x16& acc, bool hit, float tau, unsigned q, int rel_row0, int half,
                                               int n_rows, unsigned row_begin32, WaveRecord* wave_log,
                                               int& log_cnt, const FilterArgs& a) {
  const unsigned long long mask = __ballot(hit);
  const int n_hit = __builtin_popcountll(mask);
  if (log_cnt + n_hit > (int)a.wave_log_cap) {  // wave-uniform
    *a.overflow = 1u;
    return;
  }
  if (hit) {
    const int idx = log_cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const int rel = rel_row0 + 4 * half;
    uint4* dst = (uint4*)(wave_log + idx);
    dst[0] = make_uint4(q, row_begin32 + (unsigned)rel, (unsigned)(n_rows - rel), __float_as_uint(tau));
#pragma unroll
    for (int g = 0; g < 4; ++g)
      dst[1 + g] = make_uint4(__float_as_uint(acc[4 * g]), __float_as_uint(acc[4 * g + 1]),
                              __float_as_uint(acc[4 * g + 2]), __float_as_uint(acc[4 * g + 3]));
  }
  log_cnt += n_hit;
}

// Epilogue of a filter wave: move the survivors of its record log to the per-query candidate
// lists.  The log only holds the wave's own <= 64 queries, so the survivors are first counted per
// query in LDS and ONE returning global atomic per query reserves their slots — the per-query
// counters are shared by every corpus chunk (every XCD), device-scope atomics on them are slow.
// One lane handles one record at a time (five 16-byte loads in flight per lane).
template <bool INCLUSIVE>
__device__ __forceinline__ void drain_wave_log(const WaveRecord* wave_log, int log_cnt, int lane, unsigned q0,
                                               unsigned* s_cnt, unsigned* s_base, const FilterArgs& a) {
  if (log_cnt == 0) return;  // wave-uniform
  s_cnt[lane] = 0;
  // the records were written by other lanes of this wave: wait until L2 acknowledged the stores
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < log_cnt; i += 64) {
    const uint4* src = (const uint4*)(wave_log + i);
    const uint4 h = src[0];
    const float tau = __uint_as_float(h.w);
    const int rows_left = (int)h.z;
    unsigned n = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint4 v = src[1 + g];
      const float sc[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        n += ((INCLUSIVE ? (sc[e] >= tau) : (sc[e] > tau)) && (e + 8 * g) < rows_left) ? 1u : 0u;
    }
    if (n) atomicAdd(&s_cnt[h.x - q0], n);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned mine = s_cnt[lane];
  s_base[lane] = mine ? atomicAdd(&a.cand_cnt[q0 + lane], mine) : 0u;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < log_cnt; i += 64) {
    const uint4* src = (const uint4*)(wave_log + i);
    const uint4 h = src[0];
    const float tau = __uint_as_float(h.w);
    const int rows_left = (int)h.z;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint4 v = src[1 + g];
      const float sc[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if ((INCLUSIVE ? (sc[e] >= tau) : (sc[e] > tau)) && (e + 8 * g) < rows_left) {
          const unsigned pos = atomicAdd(&s_base[h.x - q0], 1u);  // LDS: next free slot of this query
          if (pos < a.cap)
            a.cand[(size_t)h.x * a.cap + pos] = make_uint2(__float_as_uint(sc[e]), h.y + (unsigned)(e + 8 * g));
          else
            *a.overflow = 1u;
        }
      }
    }
  }
}

template <int QW, bool INCLUSIVE>
__global__ __launch_bounds__(kFilterThreads) void mips_filter_f16_pipe(FilterArgs a) {
  static_assert(QW == 1 || QW == 2, "two accumulators per unit");
  // the only LDS object of the kernel (a second one makes hipcc drain vmcnt before ds_reads):
  // three stage buffers, then per wave 2 x 64 counters for the log drain
  __shared__ __attribute__((aligned(16))) char lds[3 * kStageBytes + kFilterWaves * 512];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int half = lane >> 5;

  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u;
  const unsigned rest = b >> 3;
  const unsigned qt = rest % a.n_qtiles;
  const unsigned grp = rest / a.n_qtiles;
  const long long chunk = (long long)grp * 8 + xcd;
  const long long row_begin = a.slab_row0 + chunk * (long long)a.rows_per_chunk;
  if (row_begin >= a.slab_row1) return;  // launch padding (the grid is a multiple of 8 chunks)
  long long row_end = row_begin + a.rows_per_chunk;
  if (row_end > a.slab_row1) row_end = a.slab_row1;
  const int n_rows = (int)(row_end - row_begin);
  const int nstages = (n_rows + kStageRows - 1) / kStageRows;
  const unsigned row_begin32 = (unsigned)row_begin;
  const char* chunk_base = a.xb + row_begin * kRowBytes;

  const unsigned wave_slot = blockIdx.x * kFilterWaves + wave;
  WaveRecord* wave_log = a.wave_log + (size_t)wave_slot * a.wave_log_cap;
  int log_cnt = 0;  // wave-uniform

  const unsigned q0 = qt * (kFilterWaves * QW * 32) + wave * (QW * 32);
  f16x8 qf[QW][8];
  float tau[2] = {0.f, 0.f};
#pragma unroll
  for (int blk = 0; blk < QW; ++blk) {
    const char* qrow = (const char*)a.xq + (size_t)(q0 + blk * 32 + li) * kRowBytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[blk][j] = *(const f16x8*)(qrow + (2 * j + half) * 16);
    tau[blk] = a.tau[q0 + blk * 32 + li];
  }

  unsigned rd_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rd_off[j] = li * kRowBytes + (((2 * j + half) ^ (li & 15)) << 4);

  const int dma_row = lane >> 4;
  const int dma_slot = lane & 15;
  int dma_rel[4];
  int dma_chunk_off[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    dma_rel[e] = (wave * 4 + e) * 4 + dma_row;
    dma_chunk_off[e] = (dma_slot ^ ((e * 4 + dma_row) & 15)) * 16;
  }
  auto issue_stage = [&](int s, int buf_off) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int rel = s * kStageRows + dma_rel[e];
      rel = rel < n_rows ? rel : n_rows - 1;
      const char* src = chunk_base + (long long)rel * kRowBytes + dma_chunk_off[e];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + buf_off + (wave * 4 + e) * 1024),
                                       16, 0, 0);
    }
  };
  constexpr int kSubs = kStageRows / kSubRows;  // 4 units per stage
  // rotating LDS offsets of stage s, s+1, s+2
  int off0 = 0, off1 = kStageBytes, off2 = 2 * kStageBytes;
  issue_stage(0, off0);
  if (nstages > 1) issue_stage(1, off1);
  __syncthreads();  // prologue only: both stages landed

  // A fragments live in ONE register set: k-steps 0-3 of the next unit are re-loaded right after
  // the current unit's k-steps 0-3 were consumed, k-steps 4-7 after its last MFMA.
  f16x8 af[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) af[j] = *(const f16x8*)(lds + off0 + rd_off[j]);
  f32x16 pend[2] = {{0}, {0}};
  int pend_rel0 = -1;  // no pending unit yet

  for (int s = 0; s < nstages; ++s) {
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
      if (u == 2) {
        __syncthreads();
        if (s + 2 < nstages) issue_stage(s + 2, off2);
      }
      // where the next unit's fragments live (next stage's buffer after the last unit)
      const char* nxt = (u + 1 < kSubs) ? lds + off0 + (u + 1) * kSubBytes : lds + off1;

      f32x16 cur[2] = {{0}, {0}};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int blk = 0; blk < QW; ++blk)
          cur[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], qf[blk][j], cur[blk], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *(const f16x8*)(nxt + rd_off[j]);
#pragma unroll
      for (int j = 4; j < 8; ++j) {
#pragma unroll
        for (int blk = 0; blk < QW; ++blk)
          cur[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], qf[blk][j], cur[blk], 0, 0, 0);
      }
#pragma unroll
      for (int j = 4; j < 8; ++j) af[j] = *(const f16x8*)(nxt + rd_off[j]);

      // lane-local test of the PREVIOUS unit, scheduled under the MFMAs just issued
      bool hit[2] = {false, false};
#pragma unroll
      for (int blk = 0; blk < QW; ++blk) {
        float m = pend[blk][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = __builtin_fmaxf(m, pend[blk][r]);
        hit[blk] = INCLUSIVE ? (m >= tau[blk]) : (m > tau[blk]);
      }
      if (__builtin_expect(__any((hit[0] || hit[1]) && pend_rel0 >= 0), 0)) {
#pragma unroll
        for (int blk = 0; blk < QW; ++blk)
          if (__any(hit[blk]))
            append_records(pend[blk], hit[blk], tau[blk], q0 + blk * 32 + li, pend_rel0, half, n_rows, row_begin32,
                           wave_log, log_cnt, a);
      }

      pend[0] = cur[0];
      pend[1] = cur[1];
      pend_rel0 = s * kStageRows + u * kSubRows;
    }
    const int t = off0;
    off0 = off1;
    off1 = off2;
    off2 = t;
  }
  // drain the last unit
  {
#pragma unroll
    for (int blk = 0; blk < QW; ++blk) {
      float m = pend[blk][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = __builtin_fmaxf(m, pend[blk][r]);
      const bool hit = INCLUSIVE ? (m >= tau[blk]) : (m > tau[blk]);
      if (__any(hit))
        append_records(pend[blk], hit, tau[blk], q0 + blk * 32 + li, pend_rel0, half, n_rows, row_begin32, wave_log,
                       log_cnt, a);
    }
  }
  unsigned* s_cnt = (unsigned*)(lds + 3 * kStageBytes + wave * 512);
  drain_wave_log<INCLUSIVE>(wave_log, log_cnt, lane, q0, s_cnt, s_cnt + 64, a);
}

// ---------------------------------------------------------------------------------------
// merge kernel: one workgroup per query
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMergeThreads) void topk_merge(MergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const unsigned q = blockIdx.x;
  const int tid = threadIdx.x;
  const unsigned raw_cnt = a.cand_cnt[q];
  if (raw_cnt == 0) return;  // nothing passed the threshold this round
  const unsigned cnt = raw_cnt < a.cap ? raw_cnt : a.cap;
  const unsigned nrun = a.run_n[q];
  const unsigned total = nrun + cnt;
  unsigned P = 2;
  while (P < total) P <<= 1;

  for (unsigned i = tid; i < P; i += kMergeThreads) {
    unsigned long long key = 0ull;  // below every real key
    if (i < nrun) {
      key = a.run_keys[(size_t)q * a.k + i];
    } else if (i < total) {
      const uint2 c = a.cand[(size_t)q * a.cap + (i - nrun)];
      key = pack_key(__uint_as_float(c.x), c.y);
    }
    keys[i] = key;
  }
  __syncthreads();

  if (!a.dedupe) {
    // Keys are distinct (one per corpus row), so the rank of a key = number of larger keys, and
    // the keys of rank < k ARE the new running list, already in order.  Every thread scans the
    // whole LDS array (broadcast reads, no barriers) instead of sorting it.
    const unsigned keep = total < (unsigned)a.k ? total : (unsigned)a.k;
    for (unsigned i = tid; i < total; i += kMergeThreads) {
      const unsigned long long mine = keys[i];
      unsigned rank = 0;
      for (unsigned j = 0; j < total; ++j) rank += keys[j] > mine ? 1u : 0u;
      if (rank < keep) {
        a.run_keys[(size_t)q * a.k + rank] = mine;
        if (rank == (unsigned)a.k - 1) a.tau[q] = float_from_ord((unsigned)(mine >> 32));
      }
    }
    if (tid == 0) {
      a.run_n[q] = keep;
      a.cand_cnt[q] = 0;
      a.stat_candidates[q] += raw_cnt;  // one block per query: no contention
    }
    return;
  }

  for (unsigned size = 2; size <= P; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned i = tid; i < (P >> 1); i += kMergeThreads) {
        const unsigned lo = 2 * i - (i & (stride - 1));
        const unsigned hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long x = keys[lo], y = keys[hi];
        if ((x < y) == desc) {
          keys[lo] = y;
          keys[hi] = x;
        }
      }
      __syncthreads();
    }
  }

  if (a.dedupe) {
    // overflow-safe rounds re-scan rows that may already be in the running list: drop exact
    // duplicates (adjacent after the sort).  Rare path, serial.
    if (tid == 0) {
      unsigned out = 0;
      unsigned long long prev = 0ull;
      for (unsigned i = 0; i < total && out < (unsigned)a.k; ++i) {
        const unsigned long long key = keys[i];
        if (i > 0 && key == prev) continue;
        prev = key;
        keys[out++] = key;  // out <= i, so in-place compaction is safe
      }
      a.run_n[q] = out;
      keys[P] = out;  // scratch slot (P+1 elements allocated)
    }
    __syncthreads();
    const unsigned keep = (unsigned)keys[P];
    for (unsigned i = tid; i < keep; i += kMergeThreads) a.run_keys[(size_t)q * a.k + i] = keys[i];
    if (tid == 0) {
      a.tau[q] = keep == (unsigned)a.k ? float_from_ord((unsigned)(keys[a.k - 1] >> 32)) : -__builtin_inff();
      a.cand_cnt[q] = 0;
      a.stat_candidates[q] += raw_cnt;  // one block per query: no contention
    }
    return;
  }

  const unsigned keep = total < (unsigned)a.k ? total : (unsigned)a.k;
  for (unsigned i = tid; i < keep; i += kMergeThreads) a.run_keys[(size_t)q * a.k + i] = keys[i];
  if (tid == 0) {
    a.run_n[q] = keep;
    a.tau[q] = keep == (unsigned)a.k ? float_from_ord((unsigned)(keys[a.k - 1] >> 32)) : -__builtin_inff();
    a.cand_cnt[q] = 0;
    a.stat_candidates[q] += raw_cnt;  // one block per query: no contention
  }
}

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
// pad + convert queries to the fp16 [Qpad,128] operand layout, reset per-query state
__global__ void prep_queries(const void* xq, int dtype, long long nq, long long nq_pad,
                             _Float16* xq_pad, float* tau, unsigned* cand_cnt, unsigned* run_n,
                             unsigned long long* stat, int debug_nohit) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = nq_pad * kDim;
  if (i < n) {
    const long long q = i / kDim;
    float v = 0.f;
    if (q < nq) v = dtype == PROQA_F16 ? (float)((const _Float16*)xq)[i] : ((const float*)xq)[i];
    xq_pad[i] = (_Float16)v;
  }
  if (i < nq_pad) {
    tau[i] = (i < nq && !debug_nohit) ? -__builtin_inff() : __builtin_inff();  // padded queries never emit
    cand_cnt[i] = 0;
    run_n[i] = 0;
    stat[i] = 0;
  }
}

__global__ void finalize_topk(const unsigned long long* run_keys, const unsigned* run_n, long long nq,
                              int k, long long idx_offset, float* D, long long* I) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * k) return;
  const long long q = i / k;
  const int j = (int)(i - q * k);
  if ((unsigned)j < run_n[q]) {
    const unsigned long long key = run_keys[q * k + j];
    D[i] = float_from_ord((unsigned)(key >> 32));
    I[i] = idx_offset + (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  } else {
    D[i] = -3.4028234663852886e38f;  // faiss CMin<float>::neutral()
    I[i] = -1;
  }
}

__global__ void convert_rows_f32_to_f16(const float* src, _Float16* dst, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *(const f32x4*)(src + i);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *(f16x4*)(dst + i) = o;
  } else {
    for (long long j = i; j < n; ++j) dst[j] = (_Float16)src[j];
  }
}

// Merge per-shard result lists (proqa_topk_merge_device): one workgroup per query sorts the
// n_parts*k gathered entries.  Parts must be in ascending shard order (rank order of a
// row-sharded corpus): then, for equal scores, gathered position order == global id order,
// so the position doubles as the tie-break and the 64-bit ids ride along by lookup.
__global__ __launch_bounds__(kMergeThreads) void merge_lists(const float* __restrict__ D_parts,
                                                            const long long* __restrict__ I_parts,
                                                            int n_parts, long long nq, int k,
                                                            float* __restrict__ D, long long* __restrict__ I) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const long long q = blockIdx.x;
  const int tid = threadIdx.x;
  const unsigned total = (unsigned)n_parts * (unsigned)k;
  unsigned P = 2;
  while (P < total) P <<= 1;
  for (unsigned i = tid; i < P; i += kMergeThreads) {
    unsigned long long key = 0ull;
    if (i < total) {
      const unsigned p = i / k, j = i - p * k;
      const size_t src = ((size_t)p * nq + q) * k + j;
      if (I_parts[src] >= 0) key = pack_key(D_parts[src], i);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (unsigned size = 2; size <= P; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned i = tid; i < (P >> 1); i += kMergeThreads) {
        const unsigned lo = 2 * i - (i & (stride - 1));
        const unsigned hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long x = keys[lo], y = keys[hi];
        if ((x < y) == desc) {
          keys[lo] = y;
          keys[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  for (unsigned j = tid; j < (unsigned)k; j += kMergeThreads) {
    const unsigned long long key = j < P ? keys[j] : 0ull;
    if (key != 0ull) {
      const unsigned pos = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      const unsigned p = pos / k, jj = pos - p * k;
      D[q * k + j] = float_from_ord((unsigned)(key >> 32));
      I[q * k + j] = I_parts[((size_t)p * nq + q) * k + jj];
    } else {
      D[q * k + j] = -3.4028234663852886e38f;
      I[q * k + j] = -1;
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// launch wrappers (called from mips_index.cpp)
// ---------------------------------------------------------------------------------------
hipError_t launch_filter(const FilterArgs& a, int qw, bool inclusive, unsigned grid, hipStream_t st) {
  dim3 g(grid), blk(kFilterThreads);
  if (qw == 2) {
    if (inclusive)
      hipLaunchKernelGGL((mips_filter_f16_pipe<2, true>), g, blk, 0, st, a);
    else
      hipLaunchKernelGGL((mips_filter_f16_pipe<2, false>), g, blk, 0, st, a);
  } else {
    if (inclusive)
      hipLaunchKernelGGL((mips_filter_f16_pipe<1, true>), g, blk, 0, st, a);
    else
      hipLaunchKernelGGL((mips_filter_f16_pipe<1, false>), g, blk, 0, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_merge(const MergeArgs& a, unsigned nq_pad, hipStream_t st) {
  // keys for k + cap entries rounded to a power of two, +1 scratch slot
  unsigned P = 2;
  while (P < (unsigned)a.k + a.cap) P <<= 1;
  const size_t lds = ((size_t)P + 1) * sizeof(unsigned long long);
  hipLaunchKernelGGL(topk_merge, dim3(nq_pad), dim3(kMergeThreads), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_prep_queries(const void* xq, int dtype, long long nq, long long nq_pad, void* xq_pad,
                               float* tau, unsigned* cand_cnt, unsigned* run_n, unsigned long long* stat,
                               hipStream_t st) {
  const long long n = nq_pad * kDim;
  hipLaunchKernelGGL(prep_queries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xq, dtype, nq,
                     nq_pad, (_Float16*)xq_pad, tau, cand_cnt, run_n, stat, getenv("PROQA_DEBUG_NOHIT") ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_finalize(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int k,
                           long long idx_offset, float* D, long long* I, hipStream_t st) {
  const long long n = nq * k;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(finalize_topk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, run_keys, run_n,
                     nq, k, idx_offset, D, I);
  return hipGetLastError();
}

hipError_t launch_merge_lists(const float* D_parts, const long long* I_parts, int n_parts, long long nq,
                             int k, float* D, long long* I, hipStream_t st) {
  if (nq == 0) return hipSuccess;
  unsigned P = 2;
  while (P < (unsigned)n_parts * (unsigned)k) P <<= 1;
  hipLaunchKernelGGL(merge_lists, dim3((unsigned)nq), dim3(kMergeThreads), (size_t)P * 8, st, D_parts,
                     I_parts, n_parts, nq, k, D, I);
  return hipGetLastError();
}

hipError_t launch_convert_f32_to_f16(const float* src, void* dst, long long n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const long long nthreads = (n + 3) / 4;
  hipLaunchKernelGGL(convert_rows_f32_to_f16, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                     src, (_Float16*)dst, n);
  return hipGetLastError();
}

}  // namespace proqa

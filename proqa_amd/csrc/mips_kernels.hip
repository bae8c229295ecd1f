// Exhaustive inner-product top-k over an fp16 [N,128] corpus resident in HBM.
//
// Replaces faiss.IndexFlatIP.search at /root/reference/retrieval/eval_retrieval.py:102-104.
//
// Structure (gfx950 / CDNA4, wave64):
//   mips_filter_f16   Q.P^T on MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) with the score
//                     matrix never leaving registers.  A workgroup of 8 waves keeps 8*QW*32
//                     queries as MFMA B fragments in VGPRs for its whole lifetime and streams a
//                     contiguous chunk of corpus rows through a 4 x 32 KiB LDS ring filled by
//                     LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, XOR-swizzled
//                     on the SOURCE address so ds_read_b128 is conflict-free).  Each lane owns one
//                     query column of the 32x32 accumulator, so the top-k test is lane-local: max
//                     of 16 scores vs. the query's running threshold and one wave-wide branch per
//                     32-row unit.  A lane that passes logs its whole 16-score column as one
//                     80-byte record into a list it owns in HBM: no atomics, no per-score work and
//                     no memory wait inside the MFMA loop.
//   topk_merge        one workgroup per query gathers the query's records (its own lists in every
//                     chunk + the shared spill logs), keeps the scores that pass, and selects the
//                     best k of {running list, survivors} by (score desc, row asc); publishes the
//                     k-th score as the next round's threshold.
//   The host (mips_index.cpp) runs rounds over geometrically growing corpus slabs so the
//   threshold tightens quickly and later slabs produce only a few candidates per query.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mips_kernels.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = kDim * 2;                     // 256 B per fp16 corpus row
constexpr int kStageBytes = kStageRows * kRowBytes;     // 32 KiB
constexpr int kSubRows = 32;                            // one MFMA M-tile
#ifndef PROQA_DMA_AUX
#define PROQA_DMA_AUX 2
#endif
constexpr int kDmaAux = PROQA_DMA_AUX;                  // cache policy bits of the single-reader corpus stream
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

__device__ __forceinline__ float max3_f32(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// max of four / sixteen floats as ONE asm statement each: between separate asm statements hipcc puts an `s_nop` (it does not
// know the instruction behind the string), which costs an issue slot per instruction of the threshold test
__device__ __forceinline__ float max4_f32(float a, float b, float c, float d) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}
__device__ __forceinline__ float max16_f32(const float (&v)[16]) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\t"
      "v_max3_f32 %0, %0, %4, %5\n\t"
      "v_max3_f32 %0, %0, %6, %7\n\t"
      "v_max3_f32 %0, %0, %8, %9\n\t"
      "v_max3_f32 %0, %0, %10, %11\n\t"
      "v_max3_f32 %0, %0, %12, %13\n\t"
      "v_max3_f32 %0, %0, %14, %15\n\t"
      "v_max_f32 %0, %0, %16"
      : "=&v"(r)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]),
        "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
  return r;
}

// bit i set <=> v[i] > thr: a compare into VCC and an add-with-carry per value (mask = 2 mask + carry, i = 15 .. 0), no
// branches and no temporaries.  The values must be complete in their registers (no MFMA result in flight).
__device__ __forceinline__ unsigned gt_mask16_f32(const f32x16& v, float thr) {
  unsigned mask = 0u;
#define PROQA_BIT(n) "v_cmp_gt_f32 vcc, %" #n ", %17\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
  asm volatile(PROQA_BIT(16) PROQA_BIT(15) PROQA_BIT(14) PROQA_BIT(13) PROQA_BIT(12) PROQA_BIT(11) PROQA_BIT(10) PROQA_BIT(9)
               PROQA_BIT(8) PROQA_BIT(7) PROQA_BIT(6) PROQA_BIT(5) PROQA_BIT(4) PROQA_BIT(3) PROQA_BIT(2) PROQA_BIT(1)
               : "+v"(mask)
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                 "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(thr)
               : "vcc");
#undef PROQA_BIT
  return mask;
}

__device__ __forceinline__ float max2_f32(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ size_t lane_list_index(const CandidateStore& st, unsigned chunk, unsigned q, int half) {
  return ((size_t)chunk * st.nq_pad + q) * 2 + half;
}
// The list LENGTHS are kept query-major (and the spill counters slot-major): a query's merge reads the lengths of all its
// lists -- one per chunk and half -- as one contiguous run instead of one cache line per chunk (a merge launch spent a
// quarter of its time on these scattered 4-byte loads); the scatter moves to the filter's one store per lane at its end.
__device__ __forceinline__ size_t lane_cnt_index(const CandidateStore& st, unsigned chunk, unsigned q, int half) {
  return ((size_t)q * st.n_chunks + chunk) * 2 + half;
}
__device__ __forceinline__ size_t spill_cnt_index(const CandidateStore& st, unsigned chunk, unsigned qt, unsigned wave) {
  return ((size_t)qt * kFilterWaves + wave) * st.n_chunks + chunk;
}

// ---------------------------------------------------------------------------------------
// filter kernel
// ---------------------------------------------------------------------------------------
#ifdef PROQA_FILTER_STAMPS
__device__ unsigned long long g_filter_stamps[8];   // developer build: see the end of mips_filter_f16
#endif
__device__ __forceinline__ void write_record(WaveRecord* dst, const f32x16& acc, unsigned q, unsigned row0,
                                             int rows_left, float tau) {
  uint4* d = (uint4*)dst;
  d[0] = make_uint4(q, row0, (unsigned)rows_left, __float_as_uint(tau));
#pragma unroll
  for (int g = 0; g < 4; ++g)
    d[1 + g] = make_uint4(__float_as_uint(acc[4 * g]), __float_as_uint(acc[4 * g + 1]),
                          __float_as_uint(acc[4 * g + 2]), __float_as_uint(acc[4 * g + 3]));
}

// Tiling: a workgroup = 8 waves; wave w keeps query blocks (32 queries each) w*QW .. w*QW+QW-1 of
// its query tile as MFMA B fragments in VGPRs and every wave reads the same corpus sub-tile
// (32 rows, MFMA A operand) from LDS.  The loop is arranged so that the MFMA pipe always has
// queued work from a single wave:
//   * unit = one 32-row sub-tile x QW query blocks (QW independent MFMA chains, interleaved);
//   * the A fragments live in ONE register set: k-steps 0-3 of the next unit are re-loaded right
//     after the current unit consumed them, k-steps 4-7 after its last MFMA;
//   * the lane-local max-tree + threshold test of unit g-1 sits in the same basic block as unit
//     g's MFMAs (ONE wave-wide branch per unit);
//   * four LDS stage buffers and ONE barrier per stage placed mid-stage: crossing a stage
//     boundary needs no barrier, so the fragment prefetch runs straight across it.
//     At the barrier of stage s (before its unit 2): every wave is past stage s-1, whose buffer
//     is therefore free for DMA(s+3); and every wave has retired its pieces of DMA(s+1), issued two
//     stages earlier (counted vmcnt: DMA(s+2) stays in flight), so stage s+1 is readable from unit 3 on.
//
// COMPACT (the one-pass launch of a large k over thousands of queries, where almost every unit holds a score above its
// threshold): no column records.  Every accumulator register is tested on its own (one wave-wide branch per register,
// taken for ~10 % of them at 13 000 survivors per query over 8.8M rows) and a passing score is appended as the 8-byte key the
// merge sorts -- (score, row) -- to the lane's own list in HBM: 10 x fewer bytes than the 80-byte column that carries one
// such score, and a list's consecutive 8-byte appends are combined in the XCD's L2 (the open lines of a launch's lists fit).
template <int QW, int NW, bool INCLUSIVE, bool BOUNDED, bool COMPACT = false>
__global__ __launch_bounds__(NW * 64) void mips_filter_f16(FilterArgs a) {
  static_assert(((QW == 1 || QW == 2) && NW == 8) || (QW == 4 && NW == 4), "8 waves x 32/64 queries or 4 waves x 128 queries");
  static_assert(!COMPACT || (!INCLUSIVE && !BOUNDED), "compact lists: first page, strict threshold only");
  // the only LDS object of the kernel (a second one makes hipcc drain vmcnt before ds_reads)
  __shared__ __attribute__((aligned(16))) char lds[4 * kStageBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;   // MFMA row/column owned by this lane
  const int half = lane >> 5; // which 8-wide k slice of each 16-wide k step

  // block -> (xcd, query tile, corpus chunk): the n_qtiles workgroups that stream the same
  // corpus chunk get consecutive dispatch slots on the same XCD, so the chunk is fetched from
  // HBM once and re-read from that XCD's L2.
  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u;
  const unsigned rest = b >> 3;
  const unsigned qt = rest % a.store.n_qtiles;
  const unsigned grp = rest / a.store.n_qtiles;
  const unsigned chunk = grp * 8 + xcd;
  const long long row_begin = a.slab_row0 + (long long)chunk * a.rows_per_chunk;
  if (row_begin >= a.slab_row1) return;  // launch padding (the grid is a multiple of 8 chunks)
  long long row_end = row_begin + a.rows_per_chunk;
  if (row_end > a.slab_row1) row_end = a.slab_row1;
  const int n_rows = (int)(row_end - row_begin);
  const int nstages = (n_rows + kStageRows - 1) / kStageRows;
  const unsigned row_begin32 = (unsigned)row_begin;
  const char* chunk_base = a.xb + row_begin * kRowBytes;

  // resident query fragments (MFMA B operand): lane (li, half), k-step j holds the 16-byte
  // piece 2j+half of query row q0 + blk*32 + li.
  const unsigned q0 = qt * (NW * QW * 32) + wave * (QW * 32);
  f16x8 qf[QW][8];
  float tau[QW];
  float ub[QW];              // BOUNDED (pages after the first of a k > kPageK search)
  WaveRecord* lane_list[QW];
  unsigned lane_n[QW];
#pragma unroll
  for (int blk = 0; blk < QW; ++blk) {
    const unsigned q = q0 + blk * 32 + li;
    const char* qrow = (const char*)a.xq + (size_t)q * kRowBytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[blk][j] = *(const f16x8*)(qrow + (2 * j + half) * 16);
    tau[blk] = a.tau[q];
    ub[blk] = BOUNDED ? a.ub[q] : 0.f;
    lane_n[blk] = 0u;
    lane_list[blk] = a.store.lane_log + lane_list_index(a.store, chunk, q, half) * a.store.lane_cap;
  }
  const unsigned wave_slot = (chunk * a.store.n_qtiles + qt) * kFilterWaves + wave;
  WaveRecord* spill_log = a.store.spill_log + (size_t)wave_slot * kSpillCap;
  int spill_n = 0;  // wave-uniform

  // LDS read offsets of this lane's corpus fragment (MFMA A operand): row li of a 32-row
  // sub-tile, piece 2j+half stored at slot (2j+half) ^ (li & 15).
  unsigned rd_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rd_off[j] = li * kRowBytes + (((2 * j + half) ^ (li & 15)) << 4);

  // LDS-DMA: per stage each wave issues 4 instructions of 1 KiB (4 corpus rows); lane t lands
  // at base + 16 t = (row t>>4, slot t&15) and therefore fetches piece (t&15) ^ (row&15).
  // Rows past the chunk end re-read the chunk's last row (never logged: rows_left excludes them).
  constexpr int kDmaPerWave = 32 / NW;   // 1 KiB pieces of a 32 KiB stage each wave fetches
  const int dma_row = lane >> 4;
  const int dma_slot = lane & 15;
  int dma_rel[kDmaPerWave];
  int dma_piece_off[kDmaPerWave];
#pragma unroll
  for (int e = 0; e < kDmaPerWave; ++e) {
    dma_rel[e] = (wave * kDmaPerWave + e) * 4 + dma_row;
    dma_piece_off[e] = (dma_slot ^ (dma_rel[e] & 15)) * 16;
  }
  auto issue_stage = [&](int s, int buf_off) {
#pragma unroll
    for (int e = 0; e < kDmaPerWave; ++e) {
      int rel = s * kStageRows + dma_rel[e];
      rel = rel < n_rows ? rel : n_rows - 1;
      const char* src = chunk_base + (long long)rel * kRowBytes + dma_piece_off[e];
      // QW == 1 is the single-query-tile launch: this workgroup is the only reader of its chunk, so the stream
      // is fetched non-temporally (aux 2); with several query tiles the others re-read the chunk from L2
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + buf_off + (wave * kDmaPerWave + e) * 1024),
                                       16, 0, QW == 1 ? kDmaAux : 0);
    }
  };

  constexpr int kSubs = kStageRows / kSubRows;  // 4 units per stage
  // rotating LDS offsets of stage s, s+1, s+2
  // Four stage buffers: stage s is being read, s+1 has landed, s+2 and s+3 are in flight (3 x 32 KiB per CU: what a
  // single reader needs to keep HBM busy -- bytes in flight / latency).  A barrier that publishes stage s+1 waits
  // vmcnt(kDmaPerWave) when stage s+2 was issued behind it: loads retire in order, so every older DMA piece is
  // complete once at most the kDmaPerWave youngest are outstanding (record stores in between only make the wait
  // longer, never shorter: an incomplete piece of stage s+1 implies all kDmaPerWave pieces of s+2 are outstanding too).
  auto publish = [&](bool younger_stage_in_flight) {
    if (younger_stage_in_flight) {
      // (a raw s_barrier: __syncthreads() lets hipcc drain vmcnt to 0 on account of the LDS-DMA in flight)
      if constexpr (kDmaPerWave == 4)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      dma_wait_barrier();
    }
  };
  int off0 = 0, off1 = kStageBytes, off2 = 2 * kStageBytes, off3 = 3 * kStageBytes;
  issue_stage(0, off0);
  if (nstages > 1) issue_stage(1, off1);
  if (nstages > 2) issue_stage(2, off2);
  publish(nstages > 2);  // prologue: stages 0 and 1 landed

  if ((a.flags & 1u) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);   // experiment: static priority for the younger half
  if ((a.flags & 2u) && (wave & 1)) __builtin_amdgcn_s_setprio(1);         // experiment: every other wave
  f16x8 af[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) af[j] = *(const f16x8*)(lds + off0 + rd_off[j]);
  f32x16 pend[QW];
#pragma unroll
  for (int blk = 0; blk < QW; ++blk) pend[blk] = f32x16{0};
  int pend_rel0 = -1;  // no pending unit yet

  // COMPACT: one hit = one 8-byte key appended at position n & 63 of the lane's list -- no capacity branch (a list that
  // wraps is reported at the end of the kernel by its count), no row check outside the chunk's last stage: the body of a hit
  // is ONE exec-masked region (a hit's price is its chain of dependent scalar branches, ABLATIONS R4.3)
  auto compact_test = [&](auto check_rows) {
    constexpr bool kCheckRows = decltype(check_rows)::value;
#pragma unroll
    for (int blk = 0; blk < QW; ++blk) {
      unsigned long long* list = (unsigned long long*)lane_list[blk];
#ifdef PROQA_COMPACT_FLAG128   // developer A/B build: the run-time switch to the per-register test (round 5)
      if (!kCheckRows && !(a.flags & 128u)) {
#else
      if (!kCheckRows) {
#endif
        // Every unit but the chunk's last stage: ONE wave-wide branch per query block (the column maximum), then straight-line
        // code -- the 16-bit set of the lane's scores above its threshold (32 VALU operations, no branch); a lane with ONE such
        // score (almost every hit) appends (its maximum, the row of the set bit); lanes with several take the per-register
        // loop below under a second, rarely taken, branch.  (The per-register form tests 4 quads + the registers of a passing
        // quad: ~10 dependent wave-wide branches per block and unit, which is what a hit costs: ABLATIONS R4.3, R5.12.)
        float sc16[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sc16[r] = pend[blk][r];
        const float m = max16_f32(sc16);
        if (__builtin_expect(__any(m > tau[blk]), 1)) {
          const unsigned mask = gt_mask16_f32(pend[blk], tau[blk]);
          const unsigned n = (unsigned)__builtin_popcount(mask);
          if (n == 1u) {
            const int b = __builtin_ctz(mask);
            const int rel = pend_rel0 + 4 * half + (b & 3) + 8 * (b >> 2);
            list[lane_n[blk] & (kCompactKeys - 1)] = pack_key(m, row_begin32 + (unsigned)rel);
            ++lane_n[blk];
          }
          if (__builtin_expect(__any(n > 1u), 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool h = n > 1u && ((mask >> r) & 1u);
              if (__any(h)) {
                if (h) {
                  const int rel = pend_rel0 + 4 * half + (r & 3) + 8 * (r >> 2);
                  list[lane_n[blk] & (kCompactKeys - 1)] = pack_key(pend[blk][r], row_begin32 + (unsigned)rel);
                  ++lane_n[blk];
                }
              }
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // registers 4g .. 4g+3 behind ONE pre-test of their maximum (a third of the quads pass it)
        const float m4 = max4_f32(pend[blk][4 * g], pend[blk][4 * g + 1], pend[blk][4 * g + 2], pend[blk][4 * g + 3]);
        if (__builtin_expect(__any(m4 > tau[blk]), 0)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const int rel = pend_rel0 + 4 * half + (r & 3) + 8 * (r >> 2);
            const bool h = pend[blk][r] > tau[blk] && (!kCheckRows || rel < n_rows);
            if (__builtin_expect(__any(h), 0)) {   // wave-uniform; the common case (no lane passes) must be the fall-through
              if (h) {
                list[lane_n[blk] & (kCompactKeys - 1)] = pack_key(pend[blk][r], row_begin32 + (unsigned)rel);
                ++lane_n[blk];
              }
            }
          }
        }
      }
    }
  };
  auto test_and_log = [&](bool valid) {
    if constexpr (COMPACT) {
      if (!valid) return;
      // rows past the end of the chunk (they re-read its last row) exist in the chunk's LAST stage only: the per-score
      // row check is compiled into a second copy of the test that only that stage's units take
      if (pend_rel0 >= (nstages - 1) * kStageRows)
        compact_test(std::true_type());
      else
        compact_test(std::false_type());
      return;
    }
    bool hit[QW];
    bool any_hit = false;
#pragma unroll
    for (int blk = 0; blk < QW; ++blk) {
      if (BOUNDED) {  // rows scoring above the page bound were reported by an earlier page
#pragma unroll
        for (int r = 0; r < 16; ++r) pend[blk][r] = pend[blk][r] <= ub[blk] ? pend[blk][r] : -__builtin_inff();
      }
      // v_max3_f32 by hand: fmaxf() makes hipcc quiet possible NaNs of the MFMA results first (two more VALU ops per
      // block and unit); a NaN score cannot beat a threshold either way
      float sc16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sc16[r] = pend[blk][r];
      const float m = max16_f32(sc16);
      hit[blk] = INCLUSIVE ? (m >= tau[blk]) : (m > tau[blk]);
      any_hit = any_hit || hit[blk];
    }
    if (__builtin_expect(__any(any_hit && valid), 0)) {
      // Rare path of one 32-row unit: every hit lane appends its column to the list it owns under its own exec mask; a
      // lane whose list is full falls back to the wave's shared spill log (slot = running count + rank of the lane among
      // the spilling lanes).  The full lists of all blocks are dealt with behind ONE wave-wide test per unit (a ballot +
      // branch per block measured 0.6 % / 2 % slower at 18M / 2.25M rows: a hit's price is its chain of branches)
      int rel0 = pend_rel0;
      asm volatile("" : "+v"(rel0));  // keep this arithmetic out of the MFMA loop
      const int rel = rel0 + 4 * half;
      bool spill[QW];
      bool any_spill = false;
#pragma unroll
      for (int blk = 0; blk < QW; ++blk) {
        spill[blk] = false;
        if (hit[blk]) {
          if (lane_n[blk] < a.store.lane_cap) {
            write_record(lane_list[blk] + lane_n[blk], pend[blk], q0 + blk * 32 + li, row_begin32 + (unsigned)rel, n_rows - rel, tau[blk]);
            ++lane_n[blk];
          } else {
            spill[blk] = true;
          }
        }
        any_spill = any_spill || spill[blk];
      }
      if (__builtin_expect(__any(any_spill), 0)) {
#pragma unroll
        for (int blk = 0; blk < QW; ++blk) {
          const unsigned long long mask = __ballot(spill[blk]);
          if (!mask) continue;
          const int n_spill = __builtin_popcountll(mask);
          if (spill_n + n_spill > kSpillCap) {
            *a.overflow = 1u;
          } else {
            if (spill[blk]) {
              const int idx = spill_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
              write_record(spill_log + idx, pend[blk], q0 + blk * 32 + li, row_begin32 + (unsigned)rel, n_rows - rel, tau[blk]);
            }
            spill_n += n_spill;
          }
        }
      }
    }
  };

  // A wave none of whose queries can log (padding beyond nq, queries a paged search has exhausted: threshold
  // +inf) only helps streaming the corpus: it issues its share of every stage's DMA and meets the barriers,
  // but runs no MFMA / LDS-read / test work.  With <= 32 real queries 7 of the 8 waves are such waves and
  // the launch becomes a pure HBM stream instead of an MFMA-bound scan of padding.
  bool wave_live = false;
#pragma unroll
  for (int blk = 0; blk < QW; ++blk) wave_live = wave_live || (tau[blk] < __builtin_inff());
  wave_live = __any(wave_live);
  if (!wave_live) {
    for (int s = 0; s < nstages; ++s) {
      publish(s + 2 < nstages);
      if (s + 3 < nstages) issue_stage(s + 3, off3);
      const int t = off0;
      off0 = off1;
      off1 = off2;
      off2 = off3;
      off3 = t;
    }
#pragma unroll
    for (int blk = 0; blk < QW; ++blk)
      a.store.lane_cnt[lane_cnt_index(a.store, chunk, q0 + blk * 32 + li, half)] = 0u;
    if (lane == 0) a.store.spill_cnt[spill_cnt_index(a.store, chunk, qt, wave)] = 0u;
    return;
  }

#ifdef PROQA_FILTER_STAMPS
  unsigned long long stamp_mfma = 0, stamp_test = 0, stamp_units = 0;
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
  for (int s = 0; s < nstages; ++s) {
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
      if (u == 2) {
        publish(s + 2 < nstages);  // this wave's pieces of DMA(s+1) have landed; every wave is past stage s-1
        if (s + 3 < nstages) issue_stage(s + 3, off3);
      }
      // where the next unit's fragments live (next stage's buffer after the last unit)
      const char* nxt = (u + 1 < kSubs) ? lds + off0 + (u + 1) * kSubBytes : lds + off1;

      f32x16 cur[QW];
#ifdef PROQA_FILTER_STAMPS
      const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
      for (int blk = 0; blk < QW; ++blk) cur[blk] = f32x16{0};
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

#ifdef PROQA_FILTER_STAMPS
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long st1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
#endif
      // lane-local test of the PREVIOUS unit, scheduled under the MFMAs just issued
      test_and_log(pend_rel0 >= 0);
#ifdef PROQA_FILTER_STAMPS
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long st2 = __builtin_amdgcn_s_memtime();
      stamp_mfma += st1 - st0;
      stamp_test += st2 - st1;
      ++stamp_units;
#endif

#pragma unroll
      for (int blk = 0; blk < QW; ++blk) pend[blk] = cur[blk];
      pend_rel0 = s * kStageRows + u * kSubRows;
    }
    const int t = off0;
    off0 = off1;
    off1 = off2;
    off2 = off3;
    off3 = t;
  }
  test_and_log(true);  // drain the last unit
#ifdef PROQA_FILTER_STAMPS
  if (lane == 0 && wave == 3) {   // one wave per workgroup reports: sums over its units (s_memtime ticks)
    unsigned long long* dbg = g_filter_stamps;
    atomicAdd(dbg + 0, stamp_mfma);
    atomicAdd(dbg + 1, stamp_test);
    atomicAdd(dbg + 2, stamp_units);
    atomicAdd(dbg + 3, __builtin_amdgcn_s_memtime() - stamp_t0);
    atomicAdd(dbg + 4, 1ull);
  }
#endif

#pragma unroll
  for (int blk = 0; blk < QW; ++blk) {
    if (COMPACT && lane_n[blk] > kCompactKeys) {   // the list wrapped: the launch is void (the caller searches page by page)
      *a.overflow = 1u;
      lane_n[blk] = kCompactKeys;
    }
    a.store.lane_cnt[lane_cnt_index(a.store, chunk, q0 + blk * 32 + li, half)] = lane_n[blk];
  }
  if (lane == 0) a.store.spill_cnt[spill_cnt_index(a.store, chunk, qt, wave)] = (unsigned)spill_n;
}

// ---------------------------------------------------------------------------------------
// int8 nomination scan
// ---------------------------------------------------------------------------------------
// The fp16 scan above is bound by the matrix pipe (MFMA-bound from ~300 queries per corpus pass on); v_mfma_i32_32x32x32_i8
// does the same 128-long dot products at twice the rate, exactly (i32 sums), and small batches stream half the bytes.  The
// rounds of a k <= 128 search on an fp16 index therefore scan an int8 copy of the rows:
//     u_d  = (x_d - mean_d) / c_d             c_d = max |x_d - mean_d| over the shard (per-dimension equalisation), |u_d| <= 1
//     xi_d = rint(127 u_d / f_b)              f_b = max |u| over the row's 32-ROW BLOCK (one MFMA M-tile): a block scale
//     q.x  = q.mean + (f_b / 127) sum_d w_d (xi_d + r_d)          w_d = q_d c_d,  r = 127 u / f_b - xi   (|r_d| <= 1/2)
//          = q.mean + (f_b / 127) s_q [acc + sum_d e_d xi_d + sum_d (w_d / s_q) r_d]
//                                              qi = rint(w / s_q), s_q = max |w| / 127, e = w / s_q - qi, acc = sum_d qi_d xi_d
//     |q.x - q.mean - (f_b / 127) s_q acc| <= (f_b / 127) s_q (||e|| X_b + ||w / s_q|| R)        (Cauchy-Schwarz)
//                                              X_b = max ||xi|| over the block's rows, R = max ||r|| over the shard's rows
// Centring matters: q.mean is a per-query constant that does not change the ranking, and embeddings with a large common
// component would otherwise spend the eight bits on it (scripts/dev_int8_margin.py: 256 x k nominations uncentred, 2 x k
// centred on the end-to-end test's corpus).  Block scales matter: a handful of rows of large norm (tests plant 2032 rows of
// 8 x the typical norm among 18M) stretch c_d for everyone, but only their own blocks' f_b and X_b -- with one scale for the
// shard those rows quintupled every query's margin and the lists overflowed.  A row whose exact score (the fp16 filter's
// MFMA sum) beats the running threshold tau has
//     acc > A_q G_b - B_q - E_q X_b,     A_q = (tau - q.mean - slack) / s_q,  G_b = 127 / f_b,  B_q = ||w / s_q|| R,  E_q = ||e||
// so the scan NOMINATES every row above that per-block threshold (two fused multiply-adds per 32-row unit and query block)
// and can miss none; the merge re-scores the nominated rows from the fp16 rows with the fp16 filter's own MFMA sequence --
// the same bits as the fp16 scan produces -- and everything downstream (running lists, thresholds, ties, overflow-safe path,
// shards) works on exact scores.  Nothing here depends on the accuracy of mean / c / f: the residual norms are measured.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int kRowBytesI8 = kDim;                        // 128 B per int8 corpus row
constexpr int kSubBytesI8 = kSubRows * kRowBytesI8;      // 4 KiB: one 32-row MFMA M-tile

// Plain C++ on purpose (hipcc folds it into v_max3_i32): the maximum is taken of accumulators an MFMA has just written, and
// the wait states between a matrix instruction and a VALU read of its result are the compiler's to insert -- it does not
// look into an inline-asm string (the asm max tree of the fp16 scan reads the PREVIOUS unit's accumulators; a first version
// of this kernel with that tree on the fresh ones read registers the MFMA had not written yet and lost 40 % of its hits).
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int max16_i32(const i32x16& v) {
  // a tree of depth three (five independent v_max3 first), not a chain of eight dependent ones: the two query blocks' trees
  // interleave and the wave is back at its MFMAs sooner
  const int m0 = imax(imax(v[0], v[1]), v[2]), m1 = imax(imax(v[3], v[4]), v[5]), m2 = imax(imax(v[6], v[7]), v[8]),
            m3 = imax(imax(v[9], v[10]), v[11]), m4 = imax(imax(v[12], v[13]), v[14]);
  return imax(imax(imax(m0, m1), m2), imax(imax(m3, m4), v[15]));
}

// What a lane keeps of its query for the per-block threshold T_b = A G_b - B - E X_b (see above).  A is lowered by 2^-20 of
// itself -- the roundings of the two fused multiply-adds and of G_b -- so that T_b never exceeds the exact bound.
struct LaneThreshold {
  float A, B, E;
};
__device__ __forceinline__ LaneThreshold lane_threshold(float tau, const NominateParams& p) {
  LaneThreshold t;
  t.B = p.margin_r;
  t.E = p.err_norm;
  if (!(tau > -__builtin_inff())) {
    t.A = -__builtin_inff();                      // fewer than k rows yet: every row is a candidate
  } else if (tau == __builtin_inff()) {
    t.A = __builtin_inff();                       // padding / exhausted query: never
  } else {
    const float a = (tau - p.off) * p.inv_unit;   // (a NaN from non-finite parameters compares false below: nominates nothing --
    t.A = a - __builtin_fabsf(a) * 0x1p-20f;      //  such an index is never scanned: QuantStats::nonfinite)
  }
  return t;
}
__device__ __forceinline__ float block_threshold(const LaneThreshold& t, float G, float X) {
  return __builtin_fmaf(-t.E, X, __builtin_fmaf(t.A, G, -t.B));
}

// (the threshold word of such a record is the FLOAT per-block threshold the int32 scores were tested against)
// The record of the int8 scan: {first row of the lane's column, bit i set <=> accumulator i -- row (i & 3) + 8 (i >> 2) of the
// column -- exceeds the block's threshold and lies inside the slab}.  The merge re-scores the rows from fp16 data, so the
// integer scores themselves are not kept: 8 bytes and one store instead of 80 and five.  (float)acc > thr <=> acc >
// floor(thr) for integers (|acc| < 2^24 converts exactly); thr is finite and below 2^21 on this path (the column maximum
// exceeded it) or -inf.
__device__ __forceinline__ unsigned nominee_mask(const i32x16& acc, float thr) {
  const int ti = thr < -2147483000.f ? (int)0x80000000 : (int)__builtin_floorf(thr);
  // mask = 2 mask + (acc[i] > ti), i = 15 .. 0: a compare into VCC and an add-with-carry per accumulator, no temporaries
  // (the plain C++ form costs ten more registers than the kernel's 128 allow).  The accumulators were read by the
  // column maximum before this point: no MFMA result is in flight.  (Two interleaved chains -- VCC and an SGPR pair -- and
  // v_cvt_flr_i32_f32 for the threshold: measured within noise of this form, ABLATIONS R6.8.)
  unsigned mask = 0u;
#define PROQA_BIT(n) "v_cmp_gt_i32 vcc, %" #n ", %17\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
  asm volatile(PROQA_BIT(16) PROQA_BIT(15) PROQA_BIT(14) PROQA_BIT(13) PROQA_BIT(12) PROQA_BIT(11) PROQA_BIT(10) PROQA_BIT(9)
               PROQA_BIT(8) PROQA_BIT(7) PROQA_BIT(6) PROQA_BIT(5) PROQA_BIT(4) PROQA_BIT(3) PROQA_BIT(2) PROQA_BIT(1)
               : "+v"(mask)
               : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]), "v"(acc[8]),
                 "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]), "v"(ti)
               : "vcc");
#undef PROQA_BIT
  return mask;
}
// the bits of a column whose rows lie inside the slab: rows_left = rows of the chunk from the column's first row on
__device__ __forceinline__ unsigned column_rows_mask(int rows_left) {
  unsigned valid = 0u;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = rows_left - 8 * g;
    valid |= (c >= 4 ? 0xFu : (c > 0 ? (1u << c) - 1u : 0u)) << (4 * g);
  }
  return valid;
}

// Same tiling as mips_filter_f16 -- 8 waves, wave w keeps its QW x 32 queries as MFMA B fragments for its lifetime, the
// workgroup streams a contiguous chunk of rows through LDS filled by LDS-DMA, each lane owns one query column of the 32x32
// accumulator so the test is lane-local -- with what the half-size operands change:
//   * 128 rows are 16 KiB and the LDS image of a workgroup 64 KiB: TWO workgroups per CU, i.e. four waves per SIMD; no second
//     accumulator set for a software pipeline across units (<= 128 VGPRs), the other waves of the SIMD fill the matrix pipe
//     while one examines its accumulators or logs a column;
//   * the image is two PAIRS of 128-row stages (2 x 32 KiB): the barrier at the top of pair p publishes it (every wave's DMA
//     pieces of the pair have landed: issued one pair -- eight units -- earlier; the only younger vector-memory operations of
//     the wave are its record stores, COUNTED: memory operations of a wave retire in issue order, so `vmcnt(stores since)`
//     is exact and no barrier waits for a write acknowledgement) and frees the buffers of pair p-1, which receive pair p+1
//     right behind it.  One barrier per 256 rows: a stage-wise ring with one per 128 measured 3.5 % slower (ABLATIONS R5.3);
//   * a unit (32 rows x QW x 32 queries) is 4 QW MFMAs of 32 cycles instead of 8 QW;
//   * rows are 128 B = 8 pieces of 16 B; piece p of row r sits at slot p ^ ((r >> 1) & 7) of its LDS row (XOR on the DMA
//     SOURCE address, the LDS image stays lane-linear), which keeps the four lane groups of ds_read_b128 on 16 distinct
//     16-byte bank slots each (MI355X_MICROARCH.md, LDS table);
//   * the MFMA k-order is permuted identically on both operands (piece 2j + half at k-step j); integer sums do not depend on it;
//   * the block constants (G_b, X_b) of a unit are wave-uniform: scalar loads, two fused multiply-adds per unit and query block.
// A lane whose column maximum exceeds its threshold logs ONE 8-byte record {first row of its column, 16 nominee bits}
// (nominee_mask: the merge re-scores the rows from fp16 data, the integer scores are not kept; the 80-byte column of the
// fp16 scan would be five stores and ten times the write traffic); a full list is reported through `overflow` (the round is
// then re-scanned by the fp16 overflow-safe path): no spill log, no capacity branch in the hit path.
//
// NP = pair buffers of the LDS image (NP - 1 pairs of LDS-DMA in flight).  NP = 2 (64 KiB, two workgroups per CU: 64 KB in
// flight per CU) is the shipped form for every batch size.  NP = 4 -- ONE workgroup per CU with 128 KiB of LDS and three
// pairs (96 KB) in flight, what the fp16 scan's ring holds -- was built for the HBM-bound batches (QW = 1) on the round-5
// review's reading that bytes in flight bound them; it measured 10-19 % SLOWER (the second workgroup's waves are worth more
// than the third pair: ABLATIONS R6.2) and stays behind PROQA_I8_DEEP_RING=1.
//
// SPLIT (QW = 1, at most 128 queries): with one query block per wave a batch of <= 32 queries keeps ONE wave of the
// workgroup busy -- it alone walks every 32-row unit of the chunk (fragment reads -> four dependent MFMAs -> test, in series:
// ~500 cycles per unit) while seven waves only feed the DMA stream.  Row-split launches give every wave work (measured: +3 %
// at 18M rows, +8 % at 2.25M, ABLATIONS R6.2): the a.q_blocks (1, 2, 4) query
// blocks are replicated over the 8 waves, the R = 8 / q_blocks waves of a block take every R-th unit each and append to
// the block's lists (one per query and accumulator half, as ever) through list lengths kept in LDS -- an LDS atomic per
// logged record, and records are rare where queries are few; the merge sees the same lists as from any other launch.
template <int QW, int NP, bool SPLIT = false>
__global__ __launch_bounds__(kFilterThreads, NP == 2 ? 2 : 1) void mips_filter_i8(FilterArgsI8 a) {
  static_assert(QW == 1 || QW == 2, "8 waves x 32 / 64 queries");
  static_assert(NP == 2 || NP == 4, "two pair buffers (two workgroups per CU) or four (one)");
  static_assert(!SPLIT || QW == 1, "row-split launches keep one query block per wave");
  __shared__ __attribute__((aligned(16))) char lds[2 * NP * kStageBytesI8];
  constexpr int NW = kFilterWaves;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int half = lane >> 5;

  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u;
  const unsigned rest = b >> 3;
  const unsigned qt = rest % a.store.n_qtiles;
  const unsigned grp = rest / a.store.n_qtiles;
  const unsigned chunk = grp * 8 + xcd;
  const long long row_begin = a.slab_row0 + (long long)chunk * a.rows_per_chunk;
  if (row_begin >= a.slab_row1) return;
  long long row_end = row_begin + a.rows_per_chunk;
  if (row_end > a.slab_row1) row_end = a.slab_row1;
  const int n_rows = (int)(row_end - row_begin);
  const int nstages = (n_rows + kStageRows - 1) / kStageRows;
  const int npairs = (nstages + 1) / 2;
  const unsigned row_begin32 = (unsigned)row_begin;
  const signed char* chunk_base = a.xb8 + row_begin * kRowBytesI8;

  // row-split: query block and row slice of this wave; the lengths of the shared lists live in LDS
  const unsigned n_qblk = SPLIT ? a.q_blocks : (unsigned)NW;
  const unsigned n_slices = (unsigned)NW / n_qblk;
  const unsigned qwave = SPLIT ? (unsigned)wave % n_qblk : (unsigned)wave;
  const unsigned slice = SPLIT ? (unsigned)wave / n_qblk : 0u;
  __shared__ unsigned s_list_n[SPLIT ? 4 * 64 : 1];
  if constexpr (SPLIT) {
    if (tid < 4 * 64) s_list_n[tid] = 0u;   // (visible behind the first pair's barrier, which precedes every append)
  }
  const unsigned q0 = SPLIT ? qwave * 32u : qt * (NW * QW * 32) + wave * (QW * 32);
  i32x4 qf[QW][4];
  LaneThreshold thr[QW];
  uint2* lane_list[QW];
  unsigned lane_n[QW];
  bool wave_live = false;
#pragma unroll
  for (int blk = 0; blk < QW; ++blk) {
    const unsigned q = q0 + blk * 32 + li;
    const signed char* qrow = a.xq8 + (size_t)q * kRowBytesI8;
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[blk][j] = *(const i32x4*)(qrow + (2 * j + half) * 16);
    thr[blk] = lane_threshold(a.tau[q], a.qp[q]);
    wave_live = wave_live || thr[blk].A != __builtin_inff();
    lane_n[blk] = 0u;
    lane_list[blk] = (uint2*)a.store.lane_log + lane_list_index(a.store, chunk, q, half) * a.store.lane_cap;
  }
  wave_live = __any(wave_live);
  const unsigned lane_cap = a.store.lane_cap;

  unsigned rd_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rd_off[j] = li * kRowBytesI8 + (((2 * j + half) ^ ((li >> 1) & 7)) << 4);

  constexpr int kDmaPerWave = (kStageBytesI8 / 1024) / NW;   // 2 per stage
  int dma_rel[kDmaPerWave];
  int dma_piece_off[kDmaPerWave];
#pragma unroll
  for (int e = 0; e < kDmaPerWave; ++e) {
    dma_rel[e] = (wave * kDmaPerWave + e) * 8 + (lane >> 3);
    dma_piece_off[e] = ((lane & 7) ^ ((dma_rel[e] >> 1) & 7)) * 16;
  }
  auto issue_pair = [&](int p) {   // stages 2p, 2p+1 -> the pair buffer p % NP
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = 2 * p + h;
      if (s < nstages) {
#pragma unroll
        for (int e = 0; e < kDmaPerWave; ++e) {
          int rel = s * kStageRows + dma_rel[e];
          rel = rel < n_rows ? rel : n_rows - 1;
          const signed char* src = chunk_base + (long long)rel * kRowBytesI8 + dma_piece_off[e];
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(lds + ((p & (NP - 1)) * 2 + h) * kStageBytesI8 +
                                                                                     (wave * kDmaPerWave + e) * 1024),
                                           16, 0, QW == 1 ? kDmaAux : 0);
        }
      }
    }
  };
  int st_cur = 0;   // record-store instructions of this wave since the last barrier: all younger than the awaited DMA pieces
  // `p`: the pair the barrier publishes.  Younger than its DMA pieces are this wave's record stores since the last barrier
  // and (NP = 4) its pieces of the pairs p+1, p+2 -- 2 kDmaPerWave instructions per whole pair; a chunk's last pair may hold
  // one stage.  Waiting for fewer outstanding operations than that is always safe: the tail of a chunk (the last two
  // pairs) and lists of more than four records wait a little longer than they must.
  auto publish = [&](int p) {
    const int allowed = __builtin_amdgcn_readfirstlane(st_cur);
    if (NP == 4) {
      constexpr int kPairOps = 4 * kDmaPerWave;             // instructions of two whole pairs in flight behind pair p
      if (2 * (p + 2) + 2 <= nstages) {                     // pairs p+1 and p+2 are whole
        static_assert(kPairOps == 8, "the immediates below");
        if (allowed >= 4) {
          asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else if (allowed >= 2) {
          asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else if (allowed >= 1) {
          asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
      } else if (2 * (p + 1) + 2 <= nstages) {              // pair p+1 is whole, pair p+2 short or absent
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if (allowed >= 8) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (allowed >= 6) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (allowed >= 4) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (allowed >= 3) {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else if (allowed >= 2) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else if (allowed >= 1) {
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    st_cur = 0;
  };
#pragma unroll
  for (int p = 0; p < NP - 1; ++p)
    if (p < npairs) issue_pair(p);
  if (!wave_live) {   // padding / exhausted queries only: the wave helps streaming and meets the barriers
    for (int p = 0; p < npairs; ++p) {
      publish(p);
      if (p + NP - 1 < npairs) issue_pair(p + NP - 1);
    }
  }
  // The block constants {G_b, X_b} of a pair's eight units are wave-uniform: ONE scalar load of 64 bytes per pair (hipcc
  // turns `a.blk[i]` into a vector load per unit -- a memory round trip inside every unit, and one more operation in the
  // counted vmcnt).  It is an ordinary load through a wave-uniform pointer into the constant address space, so the
  // compiler emits s_load_dwordx16 AND owns its lgkmcnt accounting (an inline-asm s_load left the wait to a hand-placed
  // s_waitcnt and the destination SGPRs to register allocation's good will: advisor finding, round 5).  Requested right
  // before the pair's barrier: scalar loads return out of order with LDS reads, so any lgkmcnt wait for a fragment read is
  // a wait for this load too -- at the barrier the wave waits anyway, mid-pair (the former place) every wave stalled for
  // the scalar round trip behind its unit 2.
  auto load_blocks = [&](int p) {
    const unsigned long long addr = (unsigned long long)(a.blk + (row_begin >> 5) + 8ll * p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)addr);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(addr >> 32));
    const unsigned long long uaddr = ((unsigned long long)hi << 32) | lo;
    return *(const __attribute__((address_space(4))) f32x16*)uaddr;
  };
  for (int p = 0; wave_live && p < npairs; ++p) {
    const f32x16 bc = load_blocks(p);
    publish(p);
    if (p + NP - 1 < npairs) issue_pair(p + NP - 1);
    const char* base = lds + (p & (NP - 1)) * 2 * kStageBytesI8;
    const int nunits = (nstages - 2 * p >= 2 ? 2 : 1) * (kStageRows / kSubRows);
    i32x4 af[4];
    if constexpr (!SPLIT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *(const i32x4*)(base + rd_off[j]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u == 4 && nunits <= 4) break;   // (the chunk's last pair may hold one stage)
      if constexpr (SPLIT) {
        // this wave's units of the pair: every n_slices-th (wave-uniform); its fragments are read on the spot -- with an
        // eighth / a quarter / half of the units per wave nothing here is on the critical path of the stream
        if (((unsigned)u & (n_slices - 1u)) != slice) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *(const i32x4*)(base + u * kSubBytesI8 + rd_off[j]);
      }
      const char* nxt = base + (u + 1 < nunits ? u + 1 : u) * kSubBytesI8;
      i32x16 acc[QW];
#pragma unroll
      for (int blk = 0; blk < QW; ++blk) acc[blk] = i32x16{0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int blk = 0; blk < QW; ++blk)
          acc[blk] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[j], qf[blk][j], acc[blk], 0, 0, 0);
      }
      if constexpr (!SPLIT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *(const i32x4*)(nxt + rd_off[j]);
      }
      bool hit[QW];
      float tb[QW];
      bool any_hit = false;
#pragma unroll
      for (int blk = 0; blk < QW; ++blk) {
        tb[blk] = block_threshold(thr[blk], bc[2 * u], bc[2 * u + 1]);
        hit[blk] = (float)max16_i32(acc[blk]) > tb[blk];   // (|acc| <= 127 * 127 * 128 < 2^24: the conversion is exact)
        any_hit = any_hit || hit[blk];
      }
      if (__builtin_expect(__any(any_hit), 0)) {
        int rel0 = (2 * p * 4 + u) * kSubRows;
        asm volatile("" : "+v"(rel0));
        const int rel = rel0 + 4 * half;
#pragma unroll
        for (int blk = 0; blk < QW; ++blk) {
          if (__any(hit[blk])) {   // (wave-uniform: exactly the regions whose store is issued are counted)
            st_cur += 1;
            if (hit[blk]) {
              // (row-split: the list is shared with the other waves of the query block -- its length is an LDS counter)
              // (inline asm: for an LDS atomic of its own hipcc first drains vmcnt -- the DMA prefetch -- because LDS-DMA
              // writes LDS too; the counters are not DMA targets)
              unsigned pos = lane_n[blk];
              if constexpr (SPLIT) {
                const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)&s_list_n[qwave * 64u + (unsigned)lane];
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(addr), "v"(1u) : "memory");
              }
              const unsigned slot = pos < lane_cap ? pos : lane_cap - 1u;   // a full list keeps counting: overflow below
              unsigned mask = nominee_mask(acc[blk], tb[blk]);
              // (scalar condition: only the chunk's last unit can reach past its rows -- hipcc turned a per-lane form of this
              // test into ~40 unconditional VALU operations in every hit path)
              // (the empty asm keeps the region a BRANCH: without it hipcc hoists and if-converts it -- the ~40 VALU instructions
              // of column_rows_mask ran at the head of every hit path, selected away by a v_cndmask at the end: ABLATIONS R6.8)
#ifdef PROQA_TAILMASK_HOISTED   // developer A/B build: the round-5 form
              if ((2 * p * 4 + u + 1) * kSubRows > n_rows) mask &= column_rows_mask(n_rows - rel);
#else
              if ((2 * p * 4 + u + 1) * kSubRows > n_rows) {
                int rows_left = n_rows - rel;
                asm volatile("" : "+v"(rows_left));   // (an opaque value born inside the region: nothing of it can be hoisted)
                mask &= column_rows_mask(rows_left);
              }
#endif
              lane_list[blk][slot] = make_uint2(row_begin32 + (unsigned)rel, mask);
              ++lane_n[blk];
            }
          }
        }
      }
    }
  }
  if constexpr (SPLIT) {
    // the lengths of the shared lists: complete once every wave of the workgroup is here.  The waves of slice 0 report their
    // query block's lists; the others zero the lists of the query blocks nobody scanned for (padding queries: the merge
    // reads every list length of the tile)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (slice == 0) {
      lane_n[0] = s_list_n[qwave * 64u + (unsigned)lane];
    } else {
      // waves n_qblk .. 7 <-> query blocks n_qblk .. 7 (every one of them exactly once)
      a.store.lane_cnt[lane_cnt_index(a.store, chunk, (unsigned)wave * 32u + li, half)] = 0u;
      return;
    }
  }
#pragma unroll
  for (int blk = 0; blk < QW; ++blk) {
    if (lane_n[blk] > lane_cap) {
      *a.overflow = 1u;
      lane_n[blk] = lane_cap;
    }
    a.store.lane_cnt[lane_cnt_index(a.store, chunk, q0 + blk * 32 + li, half)] = lane_n[blk];
  }
}

// ---- quantisation of the rows ------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// per-dimension sum / minimum / maximum of the fp16 rows, one partial per workgroup (a deterministic two-level reduction).
// 16 lanes per row (16 bytes = 8 dimensions each), four rows per wave and trip.
__global__ __launch_bounds__(256) void column_stats_partial(const _Float16* __restrict__ xb, long long n, float* __restrict__ partial,
                                                            QuantStats* __restrict__ stats) {
  __shared__ float red[3][16][kDim];   // [statistic][row slot of the workgroup][dimension]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane & 15, slot = w * 4 + (lane >> 4);
  float sum[8], lo[8], hi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sum[e] = 0.f;
    lo[e] = __builtin_inff();
    hi[e] = -__builtin_inff();
  }
  bool bad = false;
  for (long long r = (long long)blockIdx.x * 16 + slot; r < n; r += (long long)gridDim.x * 16) {
    const f16x8 v = *(const f16x8*)(xb + r * kDim + sub * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)v[e];
      bad = bad || !(__builtin_fabsf(x) < __builtin_inff());
      sum[e] += x;
      lo[e] = __builtin_fminf(lo[e], x);
      hi[e] = __builtin_fmaxf(hi[e], x);
    }
  }
  if (bad) atomicOr(&stats->nonfinite, 1u);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][slot][sub * 8 + e] = sum[e];
    red[1][slot][sub * 8 + e] = lo[e];
    red[2][slot][sub * 8 + e] = hi[e];
  }
  __syncthreads();
  if (threadIdx.x < kDim) {
    const int d = threadIdx.x;
    float s_ = 0.f, l_ = __builtin_inff(), h_ = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 16; ++t) {   // fixed order: deterministic
      s_ += red[0][t][d];
      l_ = __builtin_fminf(l_, red[1][t][d]);
      h_ = __builtin_fmaxf(h_, red[2][t][d]);
    }
    float* out = partial + (size_t)blockIdx.x * 3 * kDim;
    out[d] = s_;
    out[kDim + d] = l_;
    out[2 * kDim + d] = h_;
  }
}

// col[0..127] = mean, col[128..255] = 1 / c, col[256..383] = c, with c = max |x - mean| as the quantiser computes it
__global__ __launch_bounds__(kDim) void column_stats_finish(const float* __restrict__ partial, int groups, long long n, float* __restrict__ col,
                                                            QuantStats* __restrict__ stats) {
  const int d = threadIdx.x;
  double sum = 0.0;
  float lo = __builtin_inff(), hi = -__builtin_inff();
  for (int g = 0; g < groups; ++g) {
    const float* p = partial + (size_t)g * 3 * kDim;
    sum += (double)p[d];
    lo = __builtin_fminf(lo, p[kDim + d]);
    hi = __builtin_fmaxf(hi, p[2 * kDim + d]);
  }
  const float mean = (float)(sum / (double)(n > 0 ? n : 1));
  const float c = __builtin_fmaxf(hi - mean, mean - lo);   // rounding is monotone: = max over the rows of |fl(x - mean)|
  const bool finite = c >= 0.f && c < __builtin_inff() && mean == mean;
  const float inv = finite && c > 0.f ? 1.0f / c : 0.f;
  if (!finite || !(inv < __builtin_inff())) atomicOr(&stats->nonfinite, 1u);
  // a constant dimension (c == 0) keeps its mean -- x - mean = 0 exactly, it quantises to 0 with residual 0 and its whole
  // contribution q_d x_d sits in the query's offset q.mean
  col[d] = finite ? mean : 0.f;
  col[kDim + d] = inv;
  col[2 * kDim + d] = finite ? c : 0.f;
}

// One wave per 32-row block (16 lanes per row, four rows per trip, the block's 8 KiB held in registers): u = (x - mean) / c,
// the block scale f_b = max |u|, xi = clamp(rint(127 u / f_b)); blk[b] = {G_b = 127 / f_b, X_b = max ||xi|| of its rows}; R, Xn,
// Xf of QuantStats (all rounded up a little; maxima as non-negative float bits).  Rows past n (the shard's last block) count
// as the mean.
__global__ __launch_bounds__(256) void quantise_rows_i8(const _Float16* __restrict__ xb, long long n, const float* __restrict__ col,
                                                        signed char* __restrict__ xb8, float2* __restrict__ blk,
                                                        QuantStats* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 15, rsub = lane >> 4;
  float mu[8], inv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = col[sub * 8 + e];
    inv[e] = col[kDim + sub * 8 + e];
  }
  float max_r = 0.f, max_i = 0.f, max_x = 0.f;
  const long long n_blocks = (n + 31) / 32;
  for (long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_blocks; b += (long long)gridDim.x * 4) {
    float u[8][8];
    float amax = 0.f, sx_max = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const long long row = b * 32 + it * 4 + rsub;
      f16x8 v = {0};
      const bool live = row < n;
      if (live) v = *(const f16x8*)(xb + row * kDim + sub * 8);
      float sx = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = (float)v[e];
        u[it][e] = live ? (x - mu[e]) * inv[e] : 0.f;
        amax = __builtin_fmaxf(amax, __builtin_fabsf(u[it][e]));
        sx += live ? x * x : 0.f;
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sx += __shfl_xor(sx, off, 64);
      sx_max = sx > sx_max ? sx : sx_max;
    }
    float f = wave_max_f(amax);
    if (!(f > 0.f)) f = 1.0f;                  // every row of the block is the mean (or the block is not finite: flagged)
    const float G = 127.0f / f;
    float si_max = 0.f, sr_max = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const long long row = b * 32 + it * 4 + rsub;
      typedef signed char i8x8 __attribute__((ext_vector_type(8)));
      i8x8 o;
      float sr = 0.f, si = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = u[it][e] * G;
        const float i = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, __builtin_rintf(t)));
        const float r = t - i;
        o[e] = (signed char)(int)i;
        sr += r * r;
        si += i * i;
      }
      if (row < n) *(i8x8*)(xb8 + row * kDim + sub * 8) = o;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {   // the 16 lanes of the row
        sr += __shfl_xor(sr, off, 64);
        si += __shfl_xor(si, off, 64);
      }
      // (a NaN -- non-finite rows, flagged by column_stats -- compares false and leaves the maxima alone)
      sr_max = sr > sr_max ? sr : sr_max;
      si_max = si > si_max ? si : si_max;
    }
    const float xb_ = __builtin_sqrtf(wave_max_f(si_max)) * (1.0f + 0x1p-10f);
    if (lane == 0) blk[b] = make_float2(G, xb_);
    max_r = sr_max > max_r ? sr_max : max_r;
    max_i = si_max > max_i ? si_max : max_i;
    max_x = sx_max > max_x ? sx_max : max_x;
  }
  max_r = wave_max_f(max_r);
  max_i = wave_max_f(max_i);
  max_x = wave_max_f(max_x);
  if (lane == 0) {
    const float r = __builtin_sqrtf(max_r) * (1.0f + 0x1p-10f), i = __builtin_sqrtf(max_i) * (1.0f + 0x1p-10f),
                x = __builtin_sqrtf(max_x) * (1.0f + 0x1p-10f);
    if (r > 0.f) atomicMax(&stats->max_resid, __float_as_uint(r));
    if (i > 0.f) atomicMax(&stats->max_inorm, __float_as_uint(i));
    if (x > 0.f) atomicMax(&stats->max_xnorm, __float_as_uint(x));
  }
}

// one wave per (padded) query: int8 query and its NominateParams (see the derivation at the top of this section)
__global__ __launch_bounds__(256) void prep_queries_i8(const _Float16* __restrict__ xq, long long nq_pad, const float* __restrict__ col,
                                                       const QuantStats* __restrict__ stats, signed char* __restrict__ xq8,
                                                       NominateParams* __restrict__ qp, unsigned long long* __restrict__ stat_nom) {
  const int lane = threadIdx.x & 63;
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq_pad) return;
  const float q0 = (float)xq[q * kDim + 2 * lane], q1 = (float)xq[q * kDim + 2 * lane + 1];
  const float mu0 = col[2 * lane], mu1 = col[2 * lane + 1];
  const float w0 = q0 * col[2 * kDim + 2 * lane], w1 = q1 * col[2 * kDim + 2 * lane + 1];
  const float wmax = wave_max_f(__builtin_fmaxf(__builtin_fabsf(w0), __builtin_fabsf(w1)));
  const float s_q = wmax / 127.0f;
  const bool ok = s_q > 0.f && s_q < __builtin_inff();
  const float inv = ok ? 1.0f / s_q : 0.f;
  const float u0 = w0 * inv, u1 = w1 * inv;
  const float i0 = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, __builtin_rintf(u0)));
  const float i1 = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, __builtin_rintf(u1)));
  typedef signed char i8x2 __attribute__((ext_vector_type(2)));
  const i8x2 o = {(signed char)(int)i0, (signed char)(int)i1};
  *(i8x2*)(xq8 + q * kDim + 2 * lane) = o;
  const float e0 = u0 - i0, e1 = u1 - i1;
  const float n_u = __builtin_sqrtf(wave_sum_f(u0 * u0 + u1 * u1)) * (1.0f + 0x1p-10f);
  const float n_e = __builtin_sqrtf(wave_sum_f(e0 * e0 + e1 * e1)) * (1.0f + 0x1p-10f) + 0x1p-10f;
  const float n_q = __builtin_sqrtf(wave_sum_f(q0 * q0 + q1 * q1)) * (1.0f + 0x1p-10f);
  const float abs_off = wave_sum_f(__builtin_fabsf(q0 * mu0) + __builtin_fabsf(q1 * mu1));
  double off = (double)q0 * (double)mu0 + (double)q1 * (double)mu1;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) off += __shfl_xor(off, d, 64);
  if (lane == 0) {
    const float R = __uint_as_float(stats->max_resid) + 0x1p-10f, Xf = __uint_as_float(stats->max_xnorm);
    // score units: the fp32 accumulation of the exact score's MFMA sums and the rounding of `off` itself -- added to the
    // offset, i.e. taken off the threshold.  (No term for fp16 subnormal operands: v_mfma_f32_32x32x16_f16 multiplies them as
    // their values on gfx950 -- scripts/native/mfma_f16_subnormal_probe.cpp -- so the exact score is the sum the bound is
    // about; with the term the exact-float32 mode carries for that case, corpora of values around 1e-3 nominated every row.)
    // 2^-14 ||q|| max||x||: four times the round-to-nearest worst case of the 127 additions and of the products' own
    // roundings (2^-17 sum |q_d x_d| each, Cauchy-Schwarz) -- the matrix unit's internal accumulation is not documented as
    // round-to-nearest per step; if it truncates, the worst case doubles.  ~0.01 score units against a margin of ~2.4.
    const float delta = 0x1p-14f * n_q * Xf + 0x1p-20f * abs_off;
    NominateParams p;
    p.off = (float)off + delta + __builtin_fabsf((float)off) * 0x1p-22f;
    p.inv_unit = inv;                                          // (0 for a zero query: A = 0, every row with acc = 0 > -B passes)
    p.margin_r = n_u * R * (1.0f + 0x1p-10f) + 0x1p-6f;       // + the roundings of the threshold's own arithmetic
    p.err_norm = n_e;
    if (!(stats->nonfinite == 0u)) p.margin_r = __builtin_inff();   // never scanned (QuantStats::nonfinite); would nominate everything
    qp[q] = p;
    stat_nom[q] = 0ull;
  }
}

// ---------------------------------------------------------------------------------------
// merge kernel: one workgroup per query
// ---------------------------------------------------------------------------------------
// keys[] collects {survivors of this round} then {running list}; the workgroup sorts them in registers
// (sort_keys_desc).  Overflow-safe rounds (inclusive threshold, rows may repeat) additionally drop
// adjacent duplicates.
// exact-float32 mode: the fp16 filter must log every row whose float32 score can exceed the exact
// threshold t: it tests against t - margin (margin bounds |float32 score - fp16 score| for this
// query over all rows), lowered by a few ulps for the rounding of the subtraction itself
__device__ __forceinline__ float filter_threshold(float t, float margin) {
  if (!(t > -__builtin_inff())) return t;  // -inf: everything is a candidate
  return t - margin - __builtin_fabsf(t) * 0x1p-21f;
}

// exact-float32 mode: rows the fp16 scan nominated are collected first (nom/n_nom, LDS) and re-scored
// afterwards by groups of 8 lanes (rescore_nominated)
struct ExactCtx {
  unsigned* nom;        // LDS list of nominated shard-local rows, or nullptr in fp16 mode
  unsigned* n_nom;
};

__device__ __forceinline__ void keep_scores_regs(const uint4 (&src)[5], unsigned q, bool inclusive,
                                                 unsigned long long bound, const ExactCtx& ex,
                                                 unsigned long long* keys, unsigned* n_keys, unsigned cap) {
  const uint4 h = src[0];
  if (h.x != q) return;  // spill logs mix the wave's queries
  const float tau = __uint_as_float(h.w);
  const int rows_left = (int)h.z;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint4 v = src[1 + g];
    const float sc[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool pass = inclusive ? (sc[e] >= tau) : (sc[e] > tau);
      if (pass && (e + 8 * g) < rows_left) {
        const unsigned row = h.y + (unsigned)(e + 8 * g);
        if (ex.nom) {  // the fp16 score only nominates the row
          const unsigned pos = atomicAdd(ex.n_nom, 1u);  // LDS
          if (pos < cap) ex.nom[pos] = row;
          continue;
        }
        const unsigned long long key = pack_key(sc[e], row);
        if (key < bound) {  // paged search: ties with the bound score that were already reported
          const unsigned pos = atomicAdd(n_keys, 1u);  // LDS
          if (pos < cap) keys[pos] = key;
        }
      }
    }
  }
}

__device__ __forceinline__ void keep_scores(const WaveRecord* rec, unsigned q, bool inclusive,
                                            unsigned long long bound, const ExactCtx& ex, unsigned long long* keys,
                                            unsigned* n_keys, unsigned cap) {
  uint4 buf[5];
#pragma unroll
  for (int g = 0; g < 5; ++g) buf[g] = ((const uint4*)rec)[g];
  keep_scores_regs(buf, q, inclusive, bound, ex, keys, n_keys, cap);
}

// Append the lanes' items (pred lanes only) to an LDS array: one LDS atomic per wave, ranks by mbcnt.
template <typename T>
__device__ __forceinline__ void wave_append(bool pred, T item, T* arr, unsigned* counter, unsigned cap) {
  const unsigned long long mask = __ballot(pred);
  if (!mask) return;   // wave-uniform
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, (unsigned)__builtin_popcountll(mask));   // LDS
  base = (unsigned)__shfl((int)base, 0, 64);
  const unsigned pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  if (pred && pos < cap) arr[pos] = item;
}

// v of lane (lane ^ M), M a power of two below 64, without touching the LDS crossbar (a merge launch was bound by its
// ds_bpermute traffic: 32 waves per CU x ~180 of them each): DPP moves inside a row of 16 lanes, gfx950's
// v_permlane16_swap / v_permlane32_swap across rows and halves.  M is a template parameter: a run-time switch over the
// six forms costs more scalar branching than the exchange itself.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <unsigned M>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned x, int lane) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "lane distance");
  if constexpr (M == 1) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);    // quad_perm:[1,0,3,2]
  } else if constexpr (M == 2) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
  } else if constexpr (M == 4) {                                                       // banks 0,2 <- lane+4; banks 1,3 <- lane-4
    const int r = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);      // row_shl:4
    return (unsigned)__builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xf, 0xa, false);   // row_shr:4
  } else if constexpr (M == 8) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);   // row_ror:8
  } else if constexpr (M == 16) {
    // {even rows of both, odd rows of both}: an odd row finds its partner row in the first, an even row in the second
    const u32x2 sw = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (lane & 16) ? sw[0] : sw[1];
  } else {
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);             // {lower half twice, upper half twice}
    return (lane & 32) ? sw[0] : sw[1];
  }
}
template <unsigned M>
__device__ __forceinline__ unsigned long long xor_lane_u64(unsigned long long v, int lane) {
  const unsigned lo = xor_lane_u32<M>((unsigned)v, lane), hi = xor_lane_u32<M>((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// one compare-exchange stage between lanes M apart: every key of the thread against the partner thread's
template <unsigned M, int NK>
__device__ __forceinline__ void lane_stage(unsigned long long (&v)[NK], bool keep_max, int lane) {
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const unsigned long long o = xor_lane_u64<M>(v[j], lane);
    v[j] = ((v[j] < o) == keep_max) ? o : v[j];
  }
}

// Bitonic sort (descending) of T*NK packed keys held NK per thread by a T-thread workgroup; key (tid, j) is
// element tid*NK + j of the sequence.  Strides below NK are compare-exchanges between a thread's own registers,
// strides below 64*NK are lane exchanges inside a wave (xor_lane_u64); only the two (three for T = 512) widest strides
// cross waves and go through LDS (`xchg`, T*NK keys), i.e. three barrier pairs per sort instead of one barrier per stage.
template <int NK, int T = kMergeThreads>
__device__ __forceinline__ void sort_keys_desc(unsigned long long (&v)[NK], unsigned long long* xchg, int tid) {
  constexpr unsigned P = (unsigned)T * NK;
  const int lane = tid & 63;
  for (unsigned size = 2; size <= P; size <<= 1) {
    const unsigned top = size >> 1;   // the first (widest) stride of this merge phase
    if (top >= (unsigned)NK) {
      // partner key (tid ^ m, j), m = stride / NK: all keys of a thread share the block direction (size > stride >= NK)
      const bool desc = (((unsigned)tid * NK) & size) == 0;
      for (unsigned m = top / NK; m >= 64u; m >>= 1) {
        const bool keep_max = (((unsigned)tid & m) == 0) == desc;
#pragma unroll
        for (int j = 0; j < NK; ++j) xchg[j * T + tid] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          const unsigned long long o = xchg[j * T + (tid ^ m)];
          v[j] = ((v[j] < o) == keep_max) ? o : v[j];
        }
        __syncthreads();
      }
      const unsigned m_top = top / NK < 32u ? top / NK : 32u;   // wave-uniform
      if (m_top >= 32u) lane_stage<32, NK>(v, ((tid & 32) == 0) == desc, lane);
      if (m_top >= 16u) lane_stage<16, NK>(v, ((tid & 16) == 0) == desc, lane);
      if (m_top >= 8u) lane_stage<8, NK>(v, ((tid & 8) == 0) == desc, lane);
      if (m_top >= 4u) lane_stage<4, NK>(v, ((tid & 4) == 0) == desc, lane);
      if (m_top >= 2u) lane_stage<2, NK>(v, ((tid & 2) == 0) == desc, lane);
      lane_stage<1, NK>(v, ((tid & 1) == 0) == desc, lane);
    }
#pragma unroll
    for (int s = NK / 2; s >= 1; s >>= 1) {
      if (top >= (unsigned)s) {
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          if ((j & s) == 0) {
            const bool desc = ((((unsigned)tid * NK) + j) & size) == 0;
            const unsigned long long x = v[j], y = v[j + s];
            const bool sw = (x < y) == desc;
            v[j] = sw ? y : x;
            v[j + s] = sw ? x : y;
          }
        }
      }
    }
  }
}

// Last phase of a bitonic sort on T*NK keys that already form a bitonic sequence (first half descending, second half
// ascending): log2(T NK) compare-exchange stages instead of a whole sort.  The keys are held STRIPED: v[j] is element
// j*T + tid, so the strides >= T pair registers of one thread, the strides >= 64 below that cross waves (LDS), the rest
// are lane exchanges.  Result: element j*T + tid of the descending sequence.
template <int NK, int T = kMergeThreads>
__device__ __forceinline__ void bitonic_merge_striped_desc(unsigned long long (&v)[NK], unsigned long long* xchg, int tid) {
#pragma unroll
  for (int sj = NK / 2; sj >= 1; sj >>= 1) {       // strides sj * T
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      if ((j & sj) == 0) {
        const unsigned long long x = v[j], y = v[j + sj];
        const bool sw = x < y;
        v[j] = sw ? y : x;
        v[j + sj] = sw ? x : y;
      }
    }
  }
  for (unsigned m = T / 2; m >= 64u; m >>= 1) {   // strides 64 <= m < T: partner thread tid ^ m, same register
    const bool keep_max = ((unsigned)tid & m) == 0;
#pragma unroll
    for (int j = 0; j < NK; ++j) xchg[j * T + tid] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const unsigned long long o = xchg[j * T + (tid ^ m)];
      v[j] = ((v[j] < o) == keep_max) ? o : v[j];
    }
    __syncthreads();
  }
  const int lane = tid & 63;
  lane_stage<32, NK>(v, (tid & 32) == 0, lane);
  lane_stage<16, NK>(v, (tid & 16) == 0, lane);
  lane_stage<8, NK>(v, (tid & 8) == 0, lane);
  lane_stage<4, NK>(v, (tid & 4) == 0, lane);
  lane_stage<2, NK>(v, (tid & 2) == 0, lane);
  lane_stage<1, NK>(v, (tid & 1) == 0, lane);
}

// Sort keys[0, total) (LDS, total <= 256*NK; the tail is padded with 0, which is below every real key) and
// leave the sorted sequence in registers: v[j] = element tid*NK + j.  keys[] doubles as the exchange buffer.
template <int NK, int T = kMergeThreads>
__device__ __forceinline__ void load_and_sort(unsigned long long (&v)[NK], unsigned long long* keys, unsigned total,
                                              int tid) {
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const unsigned i = j * T + tid;   // any assignment will do: it is a sort
    v[j] = i < total ? keys[i] : 0ull;
  }
  __syncthreads();   // keys[] is free for the exchanges from here on
  sort_keys_desc<NK, T>(v, keys, tid);
}

// a record of the int8 scan (mips_filter_i8: {first row of the column, nominee bits}) -> the rows to re-score
__device__ __forceinline__ void keep_nominees(uint2 rec, const ExactCtx& ex, unsigned cap) {
  unsigned m = rec.y & 0xFFFFu;
  if (!m) return;
  unsigned pos = atomicAdd(ex.n_nom, (unsigned)__builtin_popcount(m));   // LDS
  while (m) {
    const unsigned b = (unsigned)__builtin_ctz(m);
    m &= m - 1u;
    if (pos < cap) ex.nom[pos] = rec.x + (b & 3u) + 8u * (b >> 2);
    ++pos;
  }
}

#ifdef PROQA_MERGE_STAMPS
#define PROQA_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PROQA_STAMP(i) do {} while (0)
#endif
// MODE = kMergeF16, kMergeExactF32 (exact-float32 mode) or kMergeNominatedI8 (rounds of the int8 nomination scan): the two
//        re-scoring modes are instantiations of their own -- their LDS lists and registers stay out of the fp16 kernel
// CAP = keys one merge holds: kMaxSortKeys (k <= kPageK: 8 workgroups per CU) or kBigSortKeys (big pages)
// T = threads: 256, or 512 for the largest CAP (one workgroup per CU there: the second wave per SIMD hides the latencies)
// kMergeNominatedI8 keeps its nominee list and its work queue INSIDE keys[] (upper half / lower quarter: the passing keys of a
//        round are limited to CAP / 2, more raise the overflow flag) and fits seven workgroups per CU (20.6 KiB, <= 72 VGPRs):
//        with five (28.5 KiB) the 2032 merges of a round ran as 1.6 waves of workgroups, which is what an L2-resident early
//        round's merge is bound by; a 64-VGPR form (eight per CU, rows gathered in two halves) measured slower (ABLATIONS R5.5)
constexpr int kMergeF16 = 0, kMergeExactF32 = 1, kMergeNominatedI8 = 2;
// keys the merge of a nominating round holds when k <= 128: k running keys + at most 512 passing ones (a round yields
// ~1.6 k of them), 1024 nominated rows; 8.6 KiB of LDS and <= 64 VGPRs: eight workgroups per CU
constexpr int kNominatedSortKeys = 1024;
template <int MODE, int CAP, int T = kMergeThreads>
__global__ __launch_bounds__(T, T > 512 || CAP > kBigSortKeys ? 1 : (CAP > kMidSortKeys ? (MODE ? 1 : 2) : (CAP > kMaxSortKeys || MODE == 1 ? 4 : (MODE == 2 && CAP > kNominatedSortKeys ? 7 : 8))))
void topk_merge(MergeArgs a) {
  constexpr bool EXACT = MODE == kMergeExactF32;   // nominated rows are re-scored from the float32 rows in double
  constexpr bool NOM = MODE == kMergeNominatedI8;  // ... from the fp16 rows on the fp16 filter's MFMA sequence
  constexpr bool LEAN = NOM;                       // its lists live inside keys[]
  constexpr bool RESCORE = EXACT || NOM;
  constexpr bool QLDS = NOM && CAP <= kNominatedSortKeys;   // the query row of the re-scoring in LDS (see there)
  __shared__ __attribute__((aligned(16))) char s_qrow[QLDS ? kRowBytes : 16];
  __shared__ __attribute__((aligned(16))) unsigned long long keys[CAP];
  // records to gather, (list within the pass << 4) | slot: queued so that their fetches are independent and evenly
  // spread over the threads (LDS is budgeted for 8 workgroups per CU: records beyond the queue are fetched on the spot)
  // (the one-pass merge of a large k gathers ~8 records from each of 1024 lists per pass, ~8.1 k +- 0.1 k: a queue of
  // 12288 -- with 8192 four queries in ten overflowed it by a hundred records or two, 5.5 vs 5.2 ms; big pages and
  // the one-pass search's sample round put ~2 k records of a query into one pass: 4096 -- with 1024, half of them took the
  // one-at-a-time path and the sample merge of 6980 queries 1.4-1.5 ms instead of 1.1)
  // (int8 nomination rounds log ~3 x the records of an fp16 round: 2048)
  constexpr unsigned kWorkCap = CAP > kBigSortKeys ? 12288 : (CAP > kMidSortKeys ? 4096 : (CAP > kMaxSortKeys ? 3072 : (NOM ? 2048 : 1024)));
  __shared__ unsigned short s_work_own[LEAN ? 2 : kWorkCap];
  static_assert(!LEAN || kWorkCap * sizeof(unsigned short) <= CAP * sizeof(unsigned long long) / 2, "the lean work queue sits below the nominee list");
  unsigned short* const s_work = LEAN ? (unsigned short*)keys : s_work_own;
  __shared__ unsigned s_n_keys, s_n_work;
  __shared__ unsigned long long s_dummy;   // target of the stores of lanes that have nothing to append
  // work item = (list within the pass << kSlotBits) | slot in 16 bits: 1024 lists x 64 slots, or (T = 1024) 2048 x 32
  constexpr int kSlotBits = 2 * T > 1024 ? 5 : 6;
  constexpr unsigned kSlotMask = (1u << kSlotBits) - 1u;
  static_assert(kBigLaneCap <= (1 << kSlotBits) && kOnePassLaneCap <= (1 << kSlotBits) && 2 * T <= (65536 >> kSlotBits), "work item packing");
  const unsigned lane_cap = a.store.lane_cap;
  const unsigned q = blockIdx.x;
  const int tid = threadIdx.x;
  const CandidateStore& st = a.store;
  PROQA_STAMP(0);
  ExactCtx ex = {nullptr, nullptr};
  float* xq32_lds = nullptr;
  if constexpr (EXACT) {
    // this query's float32 vector and the rows the fp16 scan nominated
    __shared__ __attribute__((aligned(16))) float s_xq32[kDim];
    __shared__ unsigned s_nom[CAP];
    __shared__ unsigned s_n_nom;
    if (tid < kDim) s_xq32[tid] = a.xq32[(size_t)q * kDim + tid];   // visible after the first barrier below
    if (tid == 0) s_n_nom = 0;
    ex.nom = s_nom;
    ex.n_nom = &s_n_nom;
    xq32_lds = s_xq32;
  }
  if constexpr (NOM) {
    __shared__ unsigned s_nom_own[LEAN ? 1 : CAP];
    __shared__ unsigned s_n_nom;
    if (tid == 0) s_n_nom = 0;   // visible after the first barrier below
    if constexpr (QLDS) {
      if (tid < kRowBytes / 16) ((uint4*)s_qrow)[tid] = ((const uint4*)((const char*)a.xq16 + (size_t)q * kRowBytes))[tid];
    }
    ex.nom = LEAN ? (unsigned*)(keys + CAP / 2) : s_nom_own;   // lean: CAP row ids in the upper half of keys[]
    ex.n_nom = &s_n_nom;
  }
  const unsigned n_lists = 2 * a.n_chunks;
  const bool inclusive = a.inclusive != 0;
  const unsigned long long bound = a.bound_keys ? a.bound_keys[q] : ~0ull;

  // the wave slot that owns q in every chunk
  const unsigned tile_q = filter_tile_queries((int)a.qw);
  const unsigned qt = q / tile_q;
  const unsigned wave = (q - qt * tile_q) / (a.qw * 32);
  const unsigned spill_stride = st.n_qtiles * kFilterWaves;                              // per chunk

  // Everything that does not depend on another load is requested up front, in ONE round trip to memory: the length of the
  // running list, its first T keys (all of it for the usual k; read whether or not they are valid yet -- the slots exist --
  // and masked below), and the list lengths / spill counters of the first pass (the only pass unless a launch has more
  // than T chunks).  The records themselves are the second, and last, dependent round trip.
  const unsigned nrun = a.run_n[q];
  unsigned long long run_pref = (unsigned)tid < (unsigned)a.k ? a.run_keys[(size_t)q * a.k + tid] : 0ull;
  // Leaping rounds (mips_index.cpp plan_leap; forward rounds of a one-page search with k <= T): the scan that fed this merge
  // tested against the score at rank j < k of the running list instead of rank k -- a.tau[q] on entry.  The merged list is the
  // exact best k of the rows seen so far iff k of its keys BEAT that threshold (every row that does is in it; rows that tie
  // with it were not logged, as ever); otherwise the round is flagged (bit 3 of its overflow word) and re-scanned by the
  // overflow-safe path.  (Padding / exhausted queries -- threshold +inf, empty list -- have nothing to verify.)  What this merge leaves
  // in a.tau[q] is the threshold of the NEXT round: rank a.next_rank, or the k-th best (0: the last round, every other search)
  const float tau_in = a.leap_check ? a.tau[q] : 0.f;
  auto fell_short = [&]() {   // (one thread per query and round)
    atomicOr(a.overflow, 8u);
    if (a.short_rounds) a.short_rounds[q] |= 1u << (a.round_bit & 31);
  };
  auto note_rank = [&](unsigned i, unsigned long long key, unsigned keep) {   // key = the merged list's key of rank i
    if (keep != (unsigned)a.k) return;
    const unsigned rank = a.next_rank ? (unsigned)a.next_rank : (unsigned)a.k;
    if (i + 1 == rank) {
      const float t = float_from_ord((unsigned)(key >> 32));
      a.tau[q] = t;
      if (a.tau_filter) a.tau_filter[q] = filter_threshold(t, a.margin[q]);
    }
    if (a.leap_check && i + 1 == (unsigned)a.k && !(float_from_ord((unsigned)(key >> 32)) > tau_in)) fell_short();
  };
  unsigned cnt_first[2], n_spill_first;
  {
    const unsigned n_here = n_lists < 2u * T ? n_lists : 2u * T;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned t = (unsigned)tid + e * T;
      cnt_first[e] = t < n_here ? st.lane_cnt[lane_cnt_index(st, t >> 1, q, (int)(t & 1))] : 0u;
    }
    // (the int8 nomination scan has no spill log: its counters are not written)
    n_spill_first = !NOM && (unsigned)tid < (n_here >> 1) ? st.spill_cnt[spill_cnt_index(st, (unsigned)tid, qt, wave)] : 0u;
  }
  run_pref = (unsigned)tid < nrun ? run_pref : 0ull;

  if (tid == 0) s_n_keys = 0;

  if (!RESCORE && a.compact) {
    // Compact lists (see mips_filter_f16<COMPACT>): every entry is a key that passed its threshold.  Per sweep of up to
    // kSweep lists: one thread per list reads its length and reserves the slots (LDS atomic; the order of the keys does
    // not matter, they are sorted below); then a wave per list copies it, eight lists in flight per wave.
    unsigned* s_cnt = (unsigned*)s_work;
    constexpr unsigned kSweep = kWorkCap * sizeof(unsigned short) / (2 * sizeof(unsigned));
    unsigned* s_off = s_cnt + kSweep;
    const unsigned ccap = kCompactKeys;
    const int lane = tid & 63, w = tid >> 6;
    __syncthreads();
    for (unsigned base = 0; base < n_lists; base += kSweep) {
      const unsigned n_here = n_lists - base < kSweep ? n_lists - base : kSweep;
      for (unsigned t = tid; t < n_here; t += T) {
        const unsigned l = base + t;
        unsigned c = st.lane_cnt[lane_cnt_index(st, l >> 1, q, (int)(l & 1))];
        c = c < ccap ? c : ccap;
        s_cnt[t] = c;
        s_off[t] = c ? atomicAdd(&s_n_keys, c) : 0u;
      }
      __syncthreads();
      for (unsigned t0 = (unsigned)w * 8; t0 < n_here; t0 += (T / 64) * 8) {
        unsigned long long v[8][2];
        unsigned c[8], off[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned t = t0 + e;
          c[e] = t < n_here ? s_cnt[t] : 0u;
          off[e] = t < n_here ? s_off[t] : 0u;
          const unsigned l = base + (t < n_here ? t : 0u);
          const unsigned long long* list = (const unsigned long long*)(st.lane_log + lane_list_index(st, l >> 1, q, (int)(l & 1)) * lane_cap);
          v[e][0] = (unsigned)lane < c[e] ? list[lane] : 0ull;
          v[e][1] = (unsigned)lane + 64 < c[e] ? list[lane + 64] : 0ull;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if ((unsigned)lane < c[e] && off[e] + lane < (unsigned)CAP) keys[off[e] + lane] = v[e][0];
          if ((unsigned)lane + 64 < c[e] && off[e] + lane + 64 < (unsigned)CAP) keys[off[e] + lane + 64] = v[e][1];
        }
      }
      __syncthreads();
    }
  } else
  for (unsigned base = 0; base < n_lists; base += 2 * T) {
    if (tid == 0) s_n_work = 0;
    __syncthreads();
  PROQA_STAMP(1);
    // list lengths of this query (chunk-major, half-minor; two lists per thread) and the spill counter of
    // one chunk per thread: three independent loads, then one work item per logged record
    const unsigned n_here = (n_lists - base) < 2u * T ? (n_lists - base) : 2u * T;
    size_t li[2];
    unsigned cnt[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned t = (unsigned)tid + e * T;
      const unsigned l = base + t;
      li[e] = lane_list_index(st, l >> 1, q, (int)(l & 1));
      cnt[e] = base == 0 ? cnt_first[e] : (t < n_here ? st.lane_cnt[lane_cnt_index(st, l >> 1, q, (int)(l & 1))] : 0u);
    }
    const unsigned spill_slot0 = ((base >> 1) * st.n_qtiles + qt) * kFilterWaves + wave;   // chunk base/2
    const unsigned n_spill = base == 0 ? n_spill_first
                                       : (!NOM && (unsigned)tid < (n_here >> 1) ? st.spill_cnt[spill_cnt_index(st, (base >> 1) + (unsigned)tid, qt, wave)] : 0u);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (cnt[e]) {
        const unsigned t = (unsigned)tid + e * T;
        const unsigned pos = atomicAdd(&s_n_work, cnt[e]);   // LDS
        for (unsigned s = 0; s < cnt[e]; ++s) {
          if (pos + s < kWorkCap)
            s_work[pos + s] = (unsigned short)((t << kSlotBits) | s);
          else if constexpr (NOM)
            keep_nominees(((const uint2*)st.lane_log)[li[e] * lane_cap + s], ex, CAP);
          else
            keep_scores(st.lane_log + li[e] * lane_cap + s, q, inclusive, bound, ex, keys, &s_n_keys, CAP);
        }
      }
    }
    // spill logs of q's wave slot (usually all empty): a wave walks the non-empty chunks its lanes found
    if constexpr (!NOM) {
      const int lane = tid & 63;
      unsigned long long live = __ballot(n_spill != 0);
      while (live) {
        const int src = __builtin_ctzll(live);
        live &= live - 1;
        const unsigned n_s = (unsigned)__shfl((int)n_spill, src, 64);
        const size_t slot_s = spill_slot0 + (size_t)((tid & ~63) + src) * spill_stride;
        for (unsigned i = lane; i < n_s; i += 64)
          keep_scores(st.spill_log + slot_s * kSpillCap + i, q, inclusive, bound, ex, keys, &s_n_keys, CAP);
      }
    }
    __syncthreads();
  PROQA_STAMP(2);
    const unsigned n_work = s_n_work < kWorkCap ? s_n_work : kWorkCap;
    if constexpr (CAP > kMaxSortKeys) {
      // big merges (thousands of records per query: pages and the one-pass search of a large k) are bound by the bytes
      // of the gather, not by its request count: one lane per record, two records per thread in flight, T records per
      // trip (the five-lanes-per-record form below moves 12 records per wave and load: 9.1 vs 6.7 ms for the 6980 merges
      // of a top-10000 search)
      for (unsigned w0 = tid; w0 < n_work; w0 += 2 * T) {
        const unsigned w1 = w0 + T;
        const bool two = w1 < n_work;
        const unsigned i0 = s_work[w0], i1 = two ? s_work[w1] : i0;
        const unsigned l0 = base + (i0 >> kSlotBits), l1 = base + (i1 >> kSlotBits);
        const uint4* r0 = (const uint4*)(st.lane_log + lane_list_index(st, l0 >> 1, q, (int)(l0 & 1)) * lane_cap + (i0 & kSlotMask));
        const uint4* r1 = (const uint4*)(st.lane_log + lane_list_index(st, l1 >> 1, q, (int)(l1 & 1)) * lane_cap + (i1 & kSlotMask));
        uint4 b0[5], b1[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) b0[g] = r0[g];
#pragma unroll
        for (int g = 0; g < 5; ++g) b1[g] = r1[g];
        keep_scores_regs(b0, q, inclusive, bound, ex, keys, &s_n_keys, CAP);
        if (two) keep_scores_regs(b1, q, inclusive, bound, ex, keys, &s_n_keys, CAP);
      }
    } else if constexpr (NOM) {
      // int8 rounds: 8-byte records, one per thread and trip (a round's ~260 records of a query: one or two trips)
      for (unsigned w = tid; w < n_work; w += T) {
        const unsigned item = s_work[w];
        const unsigned l = base + (item >> kSlotBits);
        keep_nominees(((const uint2*)st.lane_log)[lane_list_index(st, l >> 1, q, (int)(l & 1)) * lane_cap + (item & kSlotMask)], ex, CAP);
      }
    } else
    // FIVE lanes per record, one 16-byte piece each (12 records per wave and load instruction, two instructions in
    // flight): the pieces of a record -- and the records of a list, which are neighbours in the queue -- are adjacent in
    // memory, so a wave's load touches a few cache lines instead of one or two per lane (one thread per record made the
    // gather as expensive as the sort).  The header lane hands row0 / rows_left / tau to the four score lanes.
    {
      const int lane = tid & 63;
      const int grp = lane / 5, piece = lane - grp * 5;
      const bool lane_on = grp < 12;
      constexpr unsigned kPerWg = 12 * (T / 64);
      auto fetch = [&](unsigned w, bool& live) -> uint4 {
        live = lane_on && w < n_work;
        if (!live) return make_uint4(0u, 0u, 0u, 0u);
        const unsigned item = s_work[w];
        const unsigned l = base + (item >> kSlotBits);
        const uint4* r = (const uint4*)(st.lane_log + lane_list_index(st, l >> 1, q, (int)(l & 1)) * lane_cap + (item & kSlotMask));
        return r[piece];
      };
      auto keep = [&](const uint4& v, bool live) {
        const int head = lane - piece;
        const unsigned row0 = (unsigned)__shfl((int)v.y, head, 64);
        const int rows_left = __shfl((int)v.z, head, 64);
        const float tau = __uint_as_float((unsigned)__shfl((int)v.w, head, 64));
        const int g = piece - 1;
        const float sc[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        if constexpr (EXACT) {   // exact-float32 mode: the fp16 score only nominates the row
          if (!live || piece == 0) return;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if ((inclusive ? (sc[e] >= tau) : (sc[e] > tau)) && (e + 8 * g) < rows_left) {
              const unsigned pos = atomicAdd(ex.n_nom, 1u);  // LDS
              if (pos < (unsigned)CAP) ex.nom[pos] = row0 + (unsigned)(e + 8 * g);
            }
          }
        } else {
          // Branch-free per lane: the four scores of a piece are tested, the survivors of the WAVE are counted by ballot
          // and get their slots from ONE LDS atomic; a lane without a survivor stores into a dummy word.  (One atomicAdd
          // per score made hipcc wrap each in its own exec-mask / ballot / readfirstlane sequence: the launch is bound by
          // scalar-unit issue, profiles/ABLATIONS.md R3.1.)
          bool p[4];
          unsigned long long key[4];
          unsigned long long m[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            key[e] = pack_key(sc[e], row0 + (unsigned)(e + 8 * g));
            // (key < bound: paged search, ties with the bound score that were already reported)
            p[e] = live && piece != 0 && (inclusive ? (sc[e] >= tau) : (sc[e] > tau)) && (e + 8 * g) < rows_left && key[e] < bound;
            m[e] = __ballot(p[e]);
          }
          const unsigned n0 = (unsigned)__builtin_popcountll(m[0]), n1 = (unsigned)__builtin_popcountll(m[1]),
                         n2 = (unsigned)__builtin_popcountll(m[2]), n3 = (unsigned)__builtin_popcountll(m[3]);
          const unsigned total = n0 + n1 + n2 + n3;
          if (total == 0) return;   // wave-uniform
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&s_n_keys, total);   // LDS
          base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
          const unsigned off[4] = {base, base + n0, base + n0 + n1, base + n0 + n1 + n2};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned pos = off[e] + __builtin_amdgcn_mbcnt_hi((unsigned)(m[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[e], 0u));
            unsigned long long* dst = (p[e] && pos < (unsigned)CAP) ? keys + pos : &s_dummy;
            *dst = key[e];
          }
        }
      };
      // loads in flight per lane (4 x 48 records per workgroup and trip: a round's ~320 records take two trips; eight in
      // flight spill registers at 8 waves per SIMD and measured slower)
      constexpr int kDepth = 4;
      for (unsigned wb = (unsigned)(tid >> 6) * 12; wb < n_work; wb += kDepth * kPerWg) {   // wave-uniform trip count
        bool live[kDepth];
        uint4 v[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) v[d] = fetch(wb + d * kPerWg + grp, live[d]);
#pragma unroll
        for (int d = 0; d < kDepth; ++d)
          if (wb + d * kPerWg < n_work) keep(v[d], live[d]);
      }
    }
    __syncthreads();
  PROQA_STAMP(3);
  }

  if constexpr (EXACT) {
    // Re-score the nominated rows from the float32 data: 8 lanes per row, each 4 x 16 bytes of the
    // 512-byte row; products of two floats are exact in double, the 128 of them are summed in double
    // and rounded ONCE -- the correctly rounded score, independent of summation order (unlike an sgemm).
    unsigned n_nom = *ex.n_nom;
    if (n_nom > (unsigned)CAP) {  // more nominations than the list holds: overflow-safe path
      if (tid == 0) *a.overflow = 1u;
      n_nom = CAP;
    }
    const float tau_exact = a.tau[q];
    const int sub = tid & 7;
    for (unsigned c = tid >> 3; c < ((n_nom + 31u) & ~31u); c += T / 8) {
      const bool live = c < n_nom;
      const unsigned row = live ? ex.nom[c] : 0u;
      double acc = 0.0;
      if (live) {
        const f32x4* xr = (const f32x4*)(a.xb32 + (size_t)row * kDim);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 x = xr[sub + 8 * t];
          const f32x4 qv = ((const f32x4*)xq32_lds)[sub + 8 * t];
          acc += (double)x[0] * (double)qv[0];
          acc += (double)x[1] * (double)qv[1];
          acc += (double)x[2] * (double)qv[2];
          acc += (double)x[3] * (double)qv[3];
        }
      }
      acc += __shfl_xor(acc, 4, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 1, 64);
      if (live && sub == 0) {
        const float score = (float)acc;
        if (inclusive ? (score >= tau_exact) : (score > tau_exact)) {
          const unsigned long long key = pack_key(score, row);
          if (key < bound) {
            const unsigned pos = atomicAdd(&s_n_keys, 1u);
            if (pos < (unsigned)CAP) keys[pos] = key;
          }
        }
      }
    }
    __syncthreads();
  }
  if constexpr (NOM) {
    // Re-score the nominated rows from the fp16 rows: a wave takes 32 of them as the A operand of the fp16 filter's own MFMA
    // sequence (rows = A, piece 2j + half at k-step j, the query in every B column), so a row's score has the bits the fp16
    // scan gives it.  Lane (li, half) ends up with the scores of rows (r & 3) + 8 (r >> 2) + 4 half of the batch in every
    // column; column 0 hands them to lanes 0..31 through LDS, which test and append them.
    __shared__ float s_sc[T / 64][32];
    // (overflow word of a nominating round: bit 0 = a lane list of the scan was full, bit 1 = this merge's capacity, bit 2 =
    // more nominations than even the 2048-key merge holds -- the host moves an index from the 1024-key to the 2048-key merge
    // on bit 1 alone, mips_index.cpp note_nomination)
    unsigned n_nom = *ex.n_nom;
    if (n_nom > (unsigned)CAP) {  // more nominations than the list holds: overflow-safe path
      if (tid == 0) atomicOr(a.overflow, n_nom > (unsigned)kMaxSortKeys ? 6u : 2u);
      n_nom = CAP;
    }
    if (tid == 0) a.stat_nominated[q] += n_nom;
    const float tau_exact = a.tau[q];
    const int lane = tid & 63, w = tid >> 6, li = lane & 31, half = lane >> 5;
    if (n_nom) {   // workgroup-uniform
      // The query's B fragments: resident in 32 VGPRs -- or, in the eight-workgroups-per-CU form (<= 64 VGPRs), read from
      // the copy of the query row in LDS right in front of every MFMA (all lanes of a half read the same 16 bytes: a
      // broadcast).  With the fragments resident at 64 VGPRs hipcc ran the eight row pieces of a batch through ONE register
      // quad: eight dependent memory round trips per batch instead of one (what made the 64-VGPR form of R5.10 slow).
      f16x8 qf[QLDS ? 1 : 8];
      const char* qrow = (const char*)a.xq16 + (size_t)q * kRowBytes;
      if constexpr (!QLDS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = *(const f16x8*)(qrow + (2 * j + half) * 16);
      }
      for (unsigned c0 = (unsigned)w * 32; c0 < n_nom; c0 += (T / 64) * 32) {   // wave-uniform trip count
        const unsigned mine = c0 + (unsigned)li < n_nom ? c0 + (unsigned)li : c0;
        const unsigned row = ex.nom[mine];
        const char* ap = a.xb16 + (size_t)row * kRowBytes;
        f16x8 af[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) af[j] = *(const f16x8*)(ap + (2 * j + half) * 16);
        f32x16 acc = {0};
        if constexpr (QLDS) {
          unsigned qoff = (unsigned)half * 16u;
          asm volatile("" : "+v"(qoff));   // (born in the loop body: the reads below stay in it)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], *(const f16x8*)(s_qrow + qoff + 32 * j), acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], qf[j], acc, 0, 0, 0);
        }
        if (li == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s_sc[w][(r & 3) + 8 * (r >> 2) + 4 * half] = acc[r];
        }
        // (LDS operations of one wave execute in program order: the reads below see the writes of lanes 0 and 32)
        const float score = s_sc[w][li];
        const bool keep = half == 0 && c0 + (unsigned)li < n_nom && score > tau_exact;
        // (passing keys go to the LOWER half of keys[] only: the nominee list occupies the upper half)
        wave_append(keep, pack_key(score, row), keys, &s_n_keys, (unsigned)CAP / 2);
      }
    }
    __syncthreads();
    if (s_n_keys > (unsigned)CAP / 2) {   // workgroup-uniform: more passing rows than the lower half holds -- overflow-safe path
      __syncthreads();
      if (tid == 0) {
        atomicOr(a.overflow, s_n_keys > (unsigned)kMaxSortKeys / 2 ? 6u : 2u);
        s_n_keys = CAP / 2;
      }
      __syncthreads();
    }
    PROQA_STAMP(7);
  }

  const unsigned n_seen = s_n_keys;
  if (n_seen == 0) {  // nothing passed the threshold this round: the list stands -- and the threshold, unless the rank changes
    if (a.next_rank || a.leap_check) {   // (k <= T: run_pref is the whole list)
      if ((unsigned)tid < nrun) note_rank((unsigned)tid, run_pref, nrun);
      if (tid == 0 && a.leap_check && nrun < (unsigned)a.k && tau_in != __builtin_inff()) fell_short();
    }
    return;
  }
  unsigned n_cand = n_seen;
  if (n_cand + nrun > (unsigned)CAP) {  // more survivors than one LDS pass holds
    if (tid == 0) atomicOr(a.overflow, NOM ? 2u : 1u);
    n_cand = CAP - nrun;
  }
  if constexpr (CAP > kMaxSortKeys) {
    // Big pages: the running list is already sorted, so only the candidates are sorted (at most CAP/2 of them: a
    // sort of half the size) and the two sorted runs are combined by ONE bitonic merge phase -- about 40 % of the
    // compare-exchanges of sorting all CAP keys again every round.
    constexpr unsigned H = CAP / 2;
    if (!inclusive && n_cand <= H && nrun <= H) {
      __syncthreads();
      unsigned long long vc[CAP / 2 / T];
      constexpr int NKC = CAP / 2 / T;
      load_and_sort<NKC, T>(vc, keys, n_cand, tid);        // vc[j] = candidate of rank tid*NKC + j (descending, 0-padded)
      __syncthreads();
      // second half of the bitonic sequence: the candidates in ASCENDING order (rank r at H + H-1-r); first half: the running list
#pragma unroll
      for (int j = 0; j < NKC; ++j) keys[2 * H - 1 - ((unsigned)tid * NKC + j)] = vc[j];
      if ((unsigned)tid < nrun) keys[tid] = run_pref;
      for (unsigned i = T + tid; i < H; i += T) keys[i] = i < nrun ? a.run_keys[(size_t)q * a.k + i] : 0ull;
      if ((unsigned)tid >= nrun) keys[tid] = 0ull;
      __syncthreads();
      unsigned long long v[CAP / T];
      constexpr int NK = CAP / T;
#pragma unroll
      for (int j = 0; j < NK; ++j) v[j] = keys[j * T + tid];
      __syncthreads();
      bitonic_merge_striped_desc<NK, T>(v, keys, tid);
      const unsigned total = n_cand + nrun;
      const unsigned keep = total < (unsigned)a.k ? total : (unsigned)a.k;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        const unsigned i = (unsigned)j * T + tid;
        if (i < keep) a.run_keys[(size_t)q * a.k + i] = v[j];
        if (i + 1 == (unsigned)a.k && keep == (unsigned)a.k) {
          const float t = float_from_ord((unsigned)(v[j] >> 32));
          a.tau[q] = t;
          if (a.tau_filter) a.tau_filter[q] = filter_threshold(t, a.margin[q]);
        }
      }
      if (tid == 0) {
        a.run_n[q] = keep;
        a.stat_candidates[q] += n_seen;
      }
      return;
    }
  }
  if ((unsigned)tid < nrun) keys[n_cand + tid] = run_pref;
  for (unsigned i = T + tid; i < nrun; i += T) keys[n_cand + i] = a.run_keys[(size_t)q * a.k + i];
  const unsigned total = n_cand + nrun;
  __syncthreads();
  PROQA_STAMP(4);

  // sort descending; forward rounds keep the first k keys as they are, inclusive rounds (rows re-scanned
  // by the overflow-safe path may already be in the running list) drop exact duplicates first
  auto finish = [&](auto& v) {
    constexpr int NK = sizeof(v) / sizeof(v[0]);
    load_and_sort<NK, T>(v, keys, total, tid);
  PROQA_STAMP(5);
    if (!inclusive) {
      const unsigned keep = total < (unsigned)a.k ? total : (unsigned)a.k;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        const unsigned i = (unsigned)tid * NK + j;
        if (i < keep) {
          a.run_keys[(size_t)q * a.k + i] = v[j];
          note_rank(i, v[j], keep);
        }
      }
      if (tid == 0) {
        a.run_n[q] = keep;
        a.stat_candidates[q] += n_seen;
        if (a.leap_check && keep < (unsigned)a.k && tau_in != __builtin_inff()) fell_short();
      }
    } else {
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        const unsigned i = (unsigned)tid * NK + j;
        if (i < total) keys[i] = v[j];
      }
    }
  };
  if (total <= 1u * T) {
    unsigned long long v[1];
    finish(v);
  } else if (total <= 2u * T) {
    unsigned long long v[2];
    finish(v);
  } else if (total <= 4u * T) {
    unsigned long long v[4];
    finish(v);
  } else if (CAP >= 8 * T && total <= 8u * T) {   // (CAP is a compile-time constant: a 4 T merge carries no 8-key network)
    unsigned long long v[8];
    finish(v);
  } else if constexpr (CAP > 8 * T) {
    static_assert(CAP == 16 * T || CAP == 32 * T || CAP == 64 * T, "largest sort");
    if (total <= 16u * T) {
      unsigned long long v[16];
      finish(v);
    } else if constexpr (CAP > 16 * T) {
      if (total <= 32u * T) {
        unsigned long long v[32];
        finish(v);
      } else if constexpr (CAP > 32 * T) {
        unsigned long long v[64];
        finish(v);
      }
    }
  }
  PROQA_STAMP(6);
  if (!inclusive) return;
  __syncthreads();
  if (tid == 0) {  // rare path, serial
    unsigned out = 0;
    unsigned long long prev = 0ull;
    for (unsigned i = 0; i < total && out < (unsigned)a.k; ++i) {
      const unsigned long long key = keys[i];
      if (i > 0 && key == prev) continue;
      prev = key;
      keys[out++] = key;  // out <= i: in-place compaction is safe
    }
    a.run_n[q] = out;
    const float t = out == (unsigned)a.k ? float_from_ord((unsigned)(keys[a.k - 1] >> 32)) : -__builtin_inff();
    a.tau[q] = t;
    if (a.tau_filter) a.tau_filter[q] = filter_threshold(t, a.margin[q]);
    a.stat_candidates[q] += n_seen;
    s_n_keys = out;
  }
  __syncthreads();
  const unsigned keep = s_n_keys;
  for (unsigned i = tid; i < keep; i += T) a.run_keys[(size_t)q * a.k + i] = keys[i];
}


// see RescoreArgs (mips_kernels.h)
constexpr unsigned kRescoreNominees = 1024;
__global__ __launch_bounds__(256) void rescore_nominated_lists(RescoreArgs a) {
  __shared__ __attribute__((aligned(16))) char s_qrow[kRowBytes];
  __shared__ unsigned s_nom[kRescoreNominees];
  __shared__ unsigned long long s_keys[kCompactKeys];
  __shared__ unsigned s_n_nom, s_n_keys, s_cnt[64], s_off[65];
  __shared__ float s_sc[4][32];
  const unsigned g = blockIdx.x, q = blockIdx.y;
  const int tid = threadIdx.x;
  const unsigned L = a.lists_per_group;
  const unsigned l0 = g * L;
  if (tid == 0) {
    s_n_nom = 0;
    s_n_keys = 0;
  }
  if (tid < kRowBytes / 16) ((uint4*)s_qrow)[tid] = ((const uint4*)((const char*)a.xq16 + (size_t)q * kRowBytes))[tid];
  if ((unsigned)tid < L) {
    const unsigned l = l0 + (unsigned)tid;
    unsigned c = l < a.in_lists ? a.in.lane_cnt[lane_cnt_index(a.in, l >> 1, q, (int)(l & 1))] : 0u;
    s_cnt[tid] = c < a.in.lane_cap ? c : a.in.lane_cap;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned o = 0;
    for (unsigned i = 0; i < L; ++i) {
      s_off[i] = o;
      o += s_cnt[i];
    }
    s_off[L] = o;
  }
  __syncthreads();
  const unsigned n_rec = s_off[L];
  ExactCtx ex = {s_nom, &s_n_nom};
  for (unsigned w = tid; w < n_rec; w += 256) {
    unsigned i = 0;
    while (w >= s_off[i + 1]) ++i;
    const unsigned l = l0 + i;
    keep_nominees(((const uint2*)a.in.lane_log)[lane_list_index(a.in, l >> 1, q, (int)(l & 1)) * a.in.lane_cap + (w - s_off[i])], ex,
                  kRescoreNominees);
  }
  __syncthreads();
  unsigned n_nom = s_n_nom;
  if (n_nom > kRescoreNominees) {
    if (tid == 0) *a.overflow = 1u;
    n_nom = kRescoreNominees;
  }
  const float tau = a.tau[q];
  const int lane = tid & 63, w = tid >> 6, li = lane & 31, half = lane >> 5;
  for (unsigned c0 = (unsigned)w * 32; c0 < n_nom; c0 += 4 * 32) {   // wave-uniform trip count (as in topk_merge<NOMINATED>)
    const unsigned mine = c0 + (unsigned)li < n_nom ? c0 + (unsigned)li : c0;
    const unsigned row = s_nom[mine];
    const char* ap = a.xb16 + (size_t)row * kRowBytes;
    f16x8 af[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) af[j] = *(const f16x8*)(ap + (2 * j + half) * 16);
    f32x16 acc = {0};
    unsigned qoff = (unsigned)half * 16u;
    asm volatile("" : "+v"(qoff));
#pragma unroll
    for (int j = 0; j < 8; ++j)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], *(const f16x8*)(s_qrow + qoff + 32 * j), acc, 0, 0, 0);
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_sc[w][(r & 3) + 8 * (r >> 2) + 4 * half] = acc[r];
    }
    const float score = s_sc[w][li];
    const bool keep = half == 0 && c0 + (unsigned)li < n_nom && score > tau;
    wave_append(keep, pack_key(score, row), s_keys, &s_n_keys, kCompactKeys);
  }
  __syncthreads();
  unsigned n = s_n_keys;
  if (n > kCompactKeys) {
    if (tid == 0) *a.overflow = 1u;
    n = kCompactKeys;
  }
  unsigned long long* dst = (unsigned long long*)(a.out.lane_log + lane_list_index(a.out, g >> 1, q, (int)(g & 1)) * a.out.lane_cap);
  if ((unsigned)tid < n) dst[tid] = s_keys[tid];
  if (tid == 0) {
    a.out.lane_cnt[lane_cnt_index(a.out, g >> 1, q, (int)(g & 1))] = n;
    if (a.stat_nominated) atomicAdd(a.stat_nominated + q, (unsigned long long)n_nom);
  }
}

// ---------------------------------------------------------------------------------------
// bootstrap: exact top-k of the first R0 corpus rows in two launches
// ---------------------------------------------------------------------------------------
// The geometric rounds start from k rows: their first three or four launches see a few thousand rows but
// log almost every score (threshold -inf or still loose), and a dense launch is bound by its scattered
// record stores, not by its arithmetic.  The bootstrap replaces them: bootstrap_scores writes the score matrix of
// rows [0, R0) as coalesced rows S[q][0..R0), bootstrap_select picks every query's top-k from its row.
//
// bootstrap_scores: one wave per 32 rows x 32 queries.  Same MFMA, operand roles and k-step order as
// mips_filter_f16 (rows = A, queries = B, piece 2j+half at step j), so a row's score has the same bits whichever
// kernel produced it (a shard's first rows are bootstrap rows, the same rows deep inside a bigger index are not).
// The accumulator tile is transposed through LDS so that S is written in 128-byte runs.
constexpr int kBootWaves = 4;
constexpr int kBootTilesPerWave = 4;   // row tiles a wave computes with the same resident query fragments
__global__ __launch_bounds__(kBootWaves * 64) void bootstrap_scores(const char* __restrict__ xb, const void* __restrict__ xq,
                                                                    int n_rows, int s_stride, float* __restrict__ S) {
  __shared__ float tile[kBootWaves][32][33];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, half = lane >> 5;
  const unsigned q0 = blockIdx.y * 32;
  const char* bp = (const char*)xq + (size_t)(q0 + li) * kRowBytes;
  f16x8 qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = *(const f16x8*)(bp + (2 * j + half) * 16);
  // the row fragments of all the wave's tiles are requested up front: ONE memory round trip per wave instead of one per
  // tile (the launch runs two waves per SIMD: nothing else hides the latency)
  f16x8 af[kBootTilesPerWave][8];
#pragma unroll
  for (int t = 0; t < kBootTilesPerWave; ++t) {
    const int row0 = ((blockIdx.x * kBootTilesPerWave + t) * kBootWaves + wave) * 32;
    const int arow = row0 + li < n_rows ? row0 + li : n_rows - 1;
    const char* ap = xb + (size_t)arow * kRowBytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) af[t][j] = *(const f16x8*)(ap + (2 * j + half) * 16);
  }
#pragma unroll
  for (int t = 0; t < kBootTilesPerWave; ++t) {
    const int row0 = ((blockIdx.x * kBootTilesPerWave + t) * kBootWaves + wave) * 32;
    if (row0 >= n_rows) break;   // wave-uniform; no workgroup barrier below: the LDS tile is private to the wave
    f32x16 acc = {0};
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][j], qf[j], acc, 0, 0, 0);
    // lane (li, half) holds query li, rows (r&3) + 8*(r>>2) + 4*half; LDS operations of one wave execute in
    // program order, so the transposed reads below see the writes of the other lanes
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[wave][li][(r & 3) + 8 * (r >> 2) + 4 * half] = acc[r];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = 2 * i + half;
      if (row0 + li < n_rows) S[(size_t)(q0 + qq) * s_stride + row0 + li] = tile[wave][qq][li];
    }
  }
}

// bootstrap_select: one workgroup per query over S[q][0, n_rows).  Thread t packs the keys of elements
// 4t .. 4t+3, + 1024, ... (16-byte loads) and keeps the largest; the k-th largest of the 256 thread maxima bounds the k-th best key from
// below (k threads hold a key at least that large), so only the keys >= that bound -- usually between k and 3k
// of them -- are collected and sorted.  More than kMaxSortKeys survivors (an adversarial order) raise the
// overflow flag; the host then repeats the page without the bootstrap.
template <int E>   // keys per thread: n_rows <= 256 E
__global__ __launch_bounds__(kMergeThreads) void bootstrap_select(const float* __restrict__ S, int n_rows, int s_stride,
                                                                  int k, int run_stride, unsigned long long* __restrict__ run_keys,
                                                                  unsigned* __restrict__ run_n, float* __restrict__ tau,
                                                                  unsigned long long* __restrict__ stat_candidates,
                                                                  unsigned* __restrict__ overflow, int tau_rank) {
  __shared__ __attribute__((aligned(16))) unsigned long long keys[kMaxSortKeys];
  __shared__ unsigned long long s_bound;
  __shared__ unsigned s_n_keys;
  const unsigned q = blockIdx.x;
  const int tid = threadIdx.x;
  if (tau[q] == __builtin_inff()) return;   // padding query: its state stays empty
  const float* row = S + (size_t)q * s_stride;
  unsigned long long mine[E];
  unsigned long long best = 0ull;
  // (four consecutive scores per 16-byte load: n_rows and the row stride are multiples of 32)
  static_assert(E % 4 == 0, "keys per thread");
#pragma unroll
  for (int e4 = 0; e4 < E / 4; ++e4) {
    const int i0 = (e4 * kMergeThreads + tid) * 4;
    const f32x4 neg = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    const f32x4 sc4 = i0 < n_rows ? *(const f32x4*)(row + i0) : neg;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = e4 * 4 + c;
      // like the filter: -inf / NaN never qualify (nor do the padding columns of the last 32-row tile)
      mine[e] = sc4[c] > -__builtin_inff() && i0 + c < n_rows ? pack_key(sc4[c], (unsigned)(i0 + c)) : 0ull;
      best = mine[e] > best ? mine[e] : best;
    }
  }
  if (tid == 0) s_n_keys = 0;
  unsigned long long bound = 1ull;   // fewer than k rows: every valid row is kept
  if (n_rows > k) {
    unsigned long long v[1] = {best};
    sort_keys_desc<1>(v, keys, tid);
    if (tid == k - 1) s_bound = v[0] ? v[0] : 1ull;
    __syncthreads();
    bound = s_bound;
  } else {
    __syncthreads();
  }
  // survivors of this thread, then ONE append per wave: exclusive scan of the counts over the lanes
  unsigned mine_n = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) mine_n += mine[e] >= bound ? 1u : 0u;
  unsigned incl = mine_n;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = (unsigned)__shfl_up((int)incl, off, 64);
    if ((tid & 63) >= off) incl += o;
  }
  unsigned base = 0;
  if ((tid & 63) == 63) base = atomicAdd(&s_n_keys, incl);   // LDS; lane 63 holds the wave total
  unsigned pos = (unsigned)__shfl((int)base, 63, 64) + incl - mine_n;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (mine[e] >= bound) {
      if (pos < (unsigned)kMaxSortKeys) keys[pos] = mine[e];
      ++pos;
    }
  }
  __syncthreads();
  unsigned total = s_n_keys;
  if (total > (unsigned)kMaxSortKeys) {
    if (tid == 0) *overflow = 1u;
    total = kMaxSortKeys;
  }
  auto finish = [&](auto& v) {
    constexpr int NK = sizeof(v) / sizeof(v[0]);
    load_and_sort<NK>(v, keys, total, tid);
    const unsigned keep = total < (unsigned)k ? total : (unsigned)k;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const unsigned i = (unsigned)tid * NK + j;
      if (i < keep) run_keys[(size_t)q * run_stride + i] = v[j];
      // (tau_rank < k: the first round behind the bootstrap leaps -- see topk_merge)
      if (i + 1 == (unsigned)tau_rank && keep == (unsigned)k) tau[q] = float_from_ord((unsigned)(v[j] >> 32));
    }
    if (tid == 0) {
      run_n[q] = keep;
      stat_candidates[q] += total;
    }
  };
  if (total <= 1u * kMergeThreads) {
    unsigned long long v[1];
    finish(v);
  } else if (total <= 2u * kMergeThreads) {
    unsigned long long v[2];
    finish(v);
  } else if (total <= 4u * kMergeThreads) {
    unsigned long long v[4];
    finish(v);
  } else {
    unsigned long long v[8];
    finish(v);
  }
}

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
// pad + convert queries to the fp16 [Qpad,128] operand layout, reset per-query state
__global__ void prep_queries(const void* xq, int dtype, long long nq, long long nq_pad, _Float16* xq_pad,
                             float* tau, unsigned* run_n, unsigned long long* stat, const unsigned char* done,
                             int reset_stat, unsigned* inexact, int debug_nohit, unsigned* overflow, unsigned* short_rounds) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (overflow && i < kOverflowWords) overflow[i] = 0u;   // the round words of the page that follows (no memset command)
  if (short_rounds && i < nq_pad) short_rounds[i] = 0u;   // leaping rounds in which the query fell short (topk_merge)
  const long long n = nq_pad * kDim;
  if (i < n) {
    const long long q = i / kDim;
    float v = 0.f;
    if (q < nq) v = dtype == PROQA_F16 ? (float)((const _Float16*)xq)[i] : ((const float*)xq)[i];
    const _Float16 hq = (_Float16)v;
    xq_pad[i] = hq;
    if (inexact && (float)hq != v && v == v) {
      atomicAdd(inexact, 1u);
      if (__builtin_isinf((float)hq) && !__builtin_isinf(v)) atomicAdd(inexact + 1, 1u);
    }
  }
  if (i < nq_pad) {
    // padded queries (and queries a paged search has already exhausted) never log
    const bool live = i < nq && !debug_nohit && !(done && done[i]);
    // debug_nohit: a finite threshold nothing beats -- the waves stay live (full MFMA work) but never log
    tau[i] = live ? -__builtin_inff() : (debug_nohit && i < nq ? 3.0e38f : __builtin_inff());
    run_n[i] = 0;
    if (reset_stat) stat[i] = 0;
  }
}

// Block 0 also reports to the host what the search's one synchronisation needs, so that no copy command sits on the
// stream between the search and whatever is enqueued behind it: `mirror` (pinned, device-visible host memory, or NULL)
// receives the overflow words of the rounds and the candidate count summed over the queries; `status` (optional device
// word) = 1 if any overflow word is set -- the host is then going to re-scan and rewrite this result (a deferred search
// hands the word to whoever consumes the result on the stream) -- else 0.
__global__ __launch_bounds__(256) void finalize_topk(const unsigned long long* run_keys, const unsigned* run_n, long long nq,
                                                     int k, long long idx_offset, float* D, long long* I, int out_stride,
                                                     int out_offset, const unsigned* overflow, unsigned* status,
                                                     SearchMirror* mirror, const unsigned long long* stat,
                                                     const unsigned long long* stat_nominated) {
  if (blockIdx.x == 0 && (status || mirror)) {
    __shared__ unsigned long long s_part[8];
    bool any = false;
    for (int w = threadIdx.x; w < kOverflowWords; w += 256) {
      const unsigned v = overflow[w];
      if (mirror) mirror->overflow[w] = v;
      any = any || v != 0u;
    }
    const int any_all = __syncthreads_or(any ? 1 : 0);
    if (threadIdx.x == 0 && status) *status = any_all ? 1u : 0u;
    if (mirror) {
      unsigned long long c = 0, m = 0;
      for (long long i = threadIdx.x; i < nq; i += 256) {
        c += stat[i];
        if (stat_nominated) m += stat_nominated[i];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_xor(c, off, 64);
        m += __shfl_xor(m, off, 64);
      }
      if ((threadIdx.x & 63) == 0) {
        s_part[threadIdx.x >> 6] = c;
        s_part[4 + (threadIdx.x >> 6)] = m;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        mirror->candidates = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        mirror->nominated = s_part[4] + s_part[5] + s_part[6] + s_part[7];
        __threadfence_system();
      }
    }
  }
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * k) return;
  const long long q = i / k;
  const int j = (int)(i - q * k);
  const long long o = q * out_stride + out_offset + j;
  if ((unsigned)j < run_n[q]) {
    const unsigned long long key = run_keys[q * k + j];
    D[o] = float_from_ord((unsigned)(key >> 32));
    I[o] = idx_offset + (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  } else {
    D[o] = -3.4028234663852886e38f;  // faiss CMin<float>::neutral()
    I[o] = -1;
  }
}

// rows of the index by id (faiss reconstruct_batch; qa/online_sampler.py:117 gathers para_embed[I] on the host): one
// thread per 16 bytes of output.  ids outside [idx_offset, idx_offset + n) -- the -1 of a short result -- give zero rows.
// out fp16: the stored fp16 rows; out fp32: the float32 copies of an exact-float32 index, else exact upcasts.
__global__ void gather_index_rows(const _Float16* __restrict__ xb16, const float* __restrict__ xb32, long long n_index,
                                  const long long* __restrict__ ids, long long n, long long idx_offset, void* out, int out_f32) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = out_f32 ? kDim / 4 : kDim / 8;      // 16-byte pieces per output row
  if (t >= n * per_row) return;
  const long long r = t / per_row;
  const int c = (int)(t - r * per_row);
  const long long row = ids[r] - idx_offset;
  const bool live = row >= 0 && row < n_index;
  if (!out_f32) {
    f16x8 v = {0};
    if (live) v = *(const f16x8*)(xb16 + row * kDim + c * 8);
    *(f16x8*)((_Float16*)out + r * kDim + c * 8) = v;
  } else {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      if (xb32) {
        v = *(const f32x4*)(xb32 + row * kDim + c * 4);
      } else {
        const _Float16* src = xb16 + row * kDim + c * 4;
        v = f32x4{(float)src[0], (float)src[1], (float)src[2], (float)src[3]};
      }
    }
    *(f32x4*)((float*)out + r * kDim + c * 4) = v;
  }
}

// one-pass search of a large k: a query that collected fewer than `want` rows above its estimated threshold
// the queries a leaping round left short are searched again on ordinary rounds as a batch of their own (mips_index.cpp
// rescue_short_queries): their padded fp16 rows out, their result rows back in
__global__ void gather_query_rows(const uint4* __restrict__ xq_pad, const int* __restrict__ ids, int n, uint4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece per thread, kRowBytes / 16 pieces per row
  constexpr int kPieces = kRowBytes / 16;
  if (i < n * kPieces) out[i] = xq_pad[(size_t)ids[i / kPieces] * kPieces + i % kPieces];
}
__global__ void scatter_result_rows(const float* __restrict__ D_src, const long long* __restrict__ I_src, const int* __restrict__ ids,
                                    int n, int k, float* __restrict__ D, long long* __restrict__ I, int out_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * k) {
    const size_t dst = (size_t)ids[i / k] * out_stride + i % k;
    D[dst] = D_src[i];
    I[dst] = I_src[i];
  }
}

__global__ void flag_short_lists(const unsigned* run_n, long long nq, unsigned want, unsigned* flag) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq && run_n[q] < want) *flag = 1u;
}

// after a page of a k > kPageK search: the last reported key bounds the next page; a query whose
// page came back short has no more rows
__global__ void advance_page(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int page_k,
                             unsigned long long* bound_keys, float* ub, unsigned char* done, const float* margin,
                             float* ub_filter) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  if (run_n[q] == (unsigned)page_k && !done[q]) {
    const unsigned long long key = run_keys[q * page_k + page_k - 1];
    bound_keys[q] = key;
    const float u = float_from_ord((unsigned)(key >> 32));
    ub[q] = u;
    // exact-float32 mode: the filter may only mask fp16 scores whose float32 score is surely above u
    if (ub_filter) ub_filter[q] = u + margin[q] + __builtin_fabsf(u) * 0x1p-21f;
  } else {
    done[q] = 1;
  }
}

// fp32 -> fp16 with a count of the values that do not survive the round trip: the index stores
// fp16, so an fp32 input is accepted only if it is exactly representable (the reference upcasts an
// fp16 .npy to float32 before faiss; such arrays pass) unless the caller allows rounding
__global__ void convert_rows_f32_to_f16(const float* src, _Float16* dst, long long n, unsigned* inexact) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  unsigned bad = 0, big = 0;   // inexact[0]: values fp16 cannot hold exactly; inexact[1]: beyond the fp16 range
  if (i + 3 < n) {
    const f32x4 v = *(const f32x4*)(src + i);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *(f16x4*)(dst + i) = o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bad += ((float)o[e] != v[e] && v[e] == v[e]) ? 1u : 0u;
      big += (__builtin_isinf((float)o[e]) && !__builtin_isinf(v[e])) ? 1u : 0u;
    }
  } else {
    for (long long j = i; j < n; ++j) {
      const _Float16 h = (_Float16)src[j];
      dst[j] = h;
      bad += ((float)h != src[j] && src[j] == src[j]) ? 1u : 0u;
      big += (__builtin_isinf((float)h) && !__builtin_isinf(src[j])) ? 1u : 0u;
    }
  }
  if (inexact && bad) atomicAdd(inexact, bad);
  if (inexact && big) atomicAdd(inexact + 1, big);
}

// ---- exact-float32 mode helpers ----------------------------------------------------------
__global__ void upconvert_rows_f16_to_f32(const _Float16* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 7 < n) {
    const f16x8 v = *(const f16x8*)(src + i);
    f32x4 lo = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    f32x4 hi = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    *(f32x4*)(dst + i) = lo;
    *(f32x4*)(dst + i + 4) = hi;
  } else {
    for (long long j = i; j < n; ++j) dst[j] = (float)src[j];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one wave per row: max over rows of the rounding-error norm and of the rounded row's norm (both are
// non-negative floats, so their bit patterns order like unsigned integers)
__global__ __launch_bounds__(256) void row_norm_stats(const float* __restrict__ xb32, const _Float16* __restrict__ xb16,
                                                      long long n_rows, unsigned* __restrict__ norm_stats) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float x0 = xb32[row * kDim + 2 * lane], x1 = xb32[row * kDim + 2 * lane + 1];
  const float h0 = (float)xb16[row * kDim + 2 * lane], h1 = (float)xb16[row * kDim + 2 * lane + 1];
  const float d0 = x0 - h0, d1 = x1 - h1;
  // rounded up a little: the bound must not be undercut by the rounding of these sums
  const float err = __builtin_sqrtf(wave_sum(d0 * d0 + d1 * d1)) * (1.0f + 0x1p-10f);
  const float nrm = __builtin_sqrtf(wave_sum(h0 * h0 + h1 * h1)) * (1.0f + 0x1p-10f);
  if (lane == 0) {
    atomicMax(norm_stats + 0, __float_as_uint(err));
    atomicMax(norm_stats + 1, __float_as_uint(nrm));
  }
}

// one wave per query: float32 copy, and margin[q] >= |q.x - fp16(q).fp16(x)| for every row x:
//   |q.x - qh.xh| <= ||qh|| ||x - xh|| + ||q - qh|| ||xh|| + ||q - qh|| ||x - xh||   (Cauchy-Schwarz)
// plus the fp32 accumulation error of the filter's MFMA sums (<= 128 * 2^-24 * ||qh|| ||xh||)
__global__ __launch_bounds__(256) void query_margins(const void* __restrict__ xq, int dtype, long long nq, long long nq_pad,
                                                     const unsigned* __restrict__ norm_stats, float* __restrict__ xq32,
                                                     float* __restrict__ margin, const float* __restrict__ tau,
                                                     float* __restrict__ tau_filter) {
  const int lane = threadIdx.x & 63;
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq_pad) return;
  float v0 = 0.f, v1 = 0.f;
  if (q < nq) {
    const long long i = q * kDim + 2 * lane;
    if (dtype == PROQA_F16) {
      v0 = (float)((const _Float16*)xq)[i];
      v1 = (float)((const _Float16*)xq)[i + 1];
    } else {
      v0 = ((const float*)xq)[i];
      v1 = ((const float*)xq)[i + 1];
    }
  }
  xq32[q * kDim + 2 * lane] = v0;
  xq32[q * kDim + 2 * lane + 1] = v1;
  const float h0 = (float)(_Float16)v0, h1 = (float)(_Float16)v1;
  const float d0 = v0 - h0, d1 = v1 - h1;
  const float nh = __builtin_sqrtf(wave_sum(h0 * h0 + h1 * h1)) * (1.0f + 0x1p-10f);
  const float nd = __builtin_sqrtf(wave_sum(d0 * d0 + d1 * d1)) * (1.0f + 0x1p-10f);
  if (lane == 0) {
    const float E = __uint_as_float(norm_stats[0]), X = __uint_as_float(norm_stats[1]);
    // last term: should the matrix cores flush fp16 subnormal operands (|v| < 2^-14) to zero, each of
    // the 128 products loses at most 2^-14 times the other factor: <= sqrt(128) 2^-14 (||q^|| + ||x^||)
    margin[q] = (nh * E + nd * X + nd * E) * (1.0f + 0x1p-10f) + 0x1p-15f * (nh + nd) * (X + E) +
                11.32f * 0x1p-14f * (nh + X);
    tau_filter[q] = tau[q];  // -inf (live) or +inf (padding / exhausted): unchanged by the margin
  }
}

// Merge per-shard result lists (proqa_topk_merge_device): one workgroup per query sorts the
// n_parts*k gathered entries.  Parts must be in ascending shard order (rank order of a
// row-sharded corpus): then, for equal scores, gathered position order == global id order,
// so the position doubles as the tie-break and the 64-bit ids ride along by lookup.
template <int NK>
__global__ __launch_bounds__(kMergeThreads) void merge_lists(const float* __restrict__ D_parts,
                                                            const long long* __restrict__ I_parts,
                                                            int n_parts, long long nq, int k, long long stride_d, long long stride_i,
                                                            float* __restrict__ D, long long* __restrict__ I,
                                                            const unsigned* __restrict__ status_src, long long status_stride,
                                                            unsigned* __restrict__ status_host) {
  // part p's [nq, k] scores / ids start stride_d / stride_i elements after part p-1's (nq*k for dense arrays)
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // max(256, pow2 >= n_parts*k) keys
  const long long q = blockIdx.x;
  const int tid = threadIdx.x;
  if (status_host && q == 0) {   // sharded search: every part's status word, straight into pinned host memory
    for (int p = tid; p < n_parts; p += kMergeThreads) status_host[p] = status_src[(size_t)p * status_stride];
    __threadfence_system();
  }
  const unsigned total = (unsigned)n_parts * (unsigned)k;
  for (unsigned i = tid; i < total; i += kMergeThreads) {
    const unsigned p = i / k, j = i - p * k;
    const size_t off = (size_t)q * k + j;
    keys[i] = I_parts[(size_t)p * stride_i + off] >= 0 ? pack_key(D_parts[(size_t)p * stride_d + off], i) : 0ull;
  }
  __syncthreads();
  unsigned long long v[NK];
  load_and_sort<NK>(v, keys, total, tid);
#pragma unroll
  for (int jj = 0; jj < NK; ++jj) {
    const unsigned j = (unsigned)tid * NK + jj;
    if (j >= (unsigned)k) continue;
    const unsigned long long key = v[jj];
    if (key != 0ull) {
      const unsigned pos = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      const unsigned p = pos / k, sj = pos - p * k;
      D[q * k + j] = float_from_ord((unsigned)(key >> 32));
      I[q * k + j] = I_parts[(size_t)p * stride_i + (size_t)q * k + sj];
    } else {
      D[q * k + j] = -3.4028234663852886e38f;
      I[q * k + j] = -1;
    }
  }
}

// The same merge for the common small case (n_parts * k <= kRankMergeKeys, e.g. 8 shards x top-80), using that every
// part is already sorted: no sort at all.  The output rank of entry j of part p is j plus, for every other part, the
// number of its keys that order before this one -- a binary search per part (keys are unique: the gathered position is
// their low word).  One barrier after the load, every surviving entry is written straight to its rank; a single part
// degenerates to a copy.  Entries with id < 0 (a shard with fewer than k rows) sit at the tail of their part.
constexpr int kRankMergeKeys = 4096;
constexpr int kRankMergeParts = 64;
__global__ __launch_bounds__(kMergeThreads) void merge_sorted_lists(const float* __restrict__ D_parts,
                                                                    const long long* __restrict__ I_parts, int n_parts,
                                                                    long long nq, int k, long long stride_d, long long stride_i,
                                                                    float* __restrict__ D, long long* __restrict__ I,
                                                                    const unsigned* __restrict__ status_src,
                                                                    long long status_stride, unsigned* __restrict__ status_host) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // n_parts * k
  __shared__ unsigned s_n[kRankMergeParts];
  const long long q = blockIdx.x;
  const int tid = threadIdx.x;
  if (status_host && q == 0) {   // sharded search: every part's status word, straight into pinned host memory
    for (int p = tid; p < n_parts; p += kMergeThreads) status_host[p] = status_src[(size_t)p * status_stride];
    __threadfence_system();
  }
  const unsigned total = (unsigned)n_parts * (unsigned)k;
  for (unsigned i = tid; i < total; i += kMergeThreads) {
    const unsigned p = i / k, j = i - p * k;
    const size_t off = (size_t)q * k + j;
    keys[i] = I_parts[(size_t)p * stride_i + off] >= 0 ? pack_key(D_parts[(size_t)p * stride_d + off], i) : 0ull;
  }
  __syncthreads();
  if (tid < n_parts) {   // valid entries of part `tid`: the first zero key
    unsigned lo = 0, hi = (unsigned)k;
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      if (keys[(unsigned)tid * k + mid] != 0ull) lo = mid + 1; else hi = mid;
    }
    s_n[tid] = lo;
  }
  __syncthreads();
  unsigned n_valid = 0;
  for (int p = 0; p < n_parts; ++p) n_valid += s_n[p];
  for (unsigned i = tid; i < total; i += kMergeThreads) {
    const unsigned long long x = keys[i];
    if (x == 0ull) continue;
    const unsigned p = i / k, j = i - p * k;
    unsigned r = j;
    for (int pp = 0; pp < n_parts && r < (unsigned)k; ++pp) {
      if ((unsigned)pp == p) continue;
      const unsigned long long* part = keys + (unsigned)pp * k;   // descending: count the keys above x
      unsigned lo = 0, hi = s_n[pp];
      while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (part[mid] > x) lo = mid + 1; else hi = mid;
      }
      r += lo;
    }
    if (r < (unsigned)k) {
      D[q * k + r] = float_from_ord((unsigned)(x >> 32));
      I[q * k + r] = I_parts[(size_t)p * stride_i + (size_t)q * k + j];
    }
  }
  for (unsigned j = n_valid + tid; j < (unsigned)k; j += kMergeThreads) {
    D[q * k + j] = -3.4028234663852886e38f;
    I[q * k + j] = -1;
  }
}

__global__ void copy_status_words(const unsigned* __restrict__ src, long long stride, int n, unsigned* __restrict__ host) {
  for (int p = threadIdx.x; p < n; p += blockDim.x) host[p] = src[(size_t)p * stride];
  __threadfence_system();
}

// Large merges (n_parts*k keys do not fit one workgroup's LDS, e.g. k = 10000 x 8 shards for
// retrieval/trec_process.py:76): the same packed keys, laid out [query][part*k + j] in HBM, are sorted
// per query by a segmented radix sort; the first k of every segment are the answer.
__global__ void pack_part_keys(const float* __restrict__ D_parts, const long long* __restrict__ I_parts,
                               int n_parts, long long nq, int k, long long stride_d, long long stride_i, long long q0, long long nq_chunk,
                               unsigned long long* __restrict__ keys) {
  const long long per_q = (long long)n_parts * k;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq_chunk * per_q) return;
  const long long qc = t / per_q;
  const unsigned i = (unsigned)(t - qc * per_q);
  const unsigned p = i / k, j = i - p * k;
  const size_t off = (size_t)(q0 + qc) * k + j;
  keys[t] = I_parts[(size_t)p * stride_i + off] >= 0 ? pack_key(D_parts[(size_t)p * stride_d + off], i) : 0ull;
}

__global__ void segment_offsets(long long n_segments, int per_segment, int* __restrict__ offsets) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= n_segments) offsets[t] = (int)(t * per_segment);
}

__global__ void emit_sorted_prefix(const unsigned long long* __restrict__ keys, const long long* __restrict__ I_parts,
                                   int n_parts, long long nq, int k, long long stride_d, long long stride_i, long long q0, long long nq_chunk,
                                   float* __restrict__ D, long long* __restrict__ I) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq_chunk * k) return;
  const long long qc = t / k;
  const int j = (int)(t - qc * k);
  const unsigned long long key = keys[qc * (long long)n_parts * k + j];
  const long long q = q0 + qc;
  if (key != 0ull) {
    const unsigned pos = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const unsigned p = pos / k, jj = pos - p * k;
    D[q * k + j] = float_from_ord((unsigned)(key >> 32));
    I[q * k + j] = I_parts[(size_t)p * stride_i + (size_t)q * k + jj];
  } else {
    D[q * k + j] = -3.4028234663852886e38f;
    I[q * k + j] = -1;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// launch wrappers (called from mips_index.cpp)
// ---------------------------------------------------------------------------------------
template <int QW, int NW, bool INCLUSIVE>
static void launch_filter_bounded(const FilterArgs& a, dim3 g, hipStream_t st) {
  if (a.ub)
    hipLaunchKernelGGL((mips_filter_f16<QW, NW, INCLUSIVE, true>), g, dim3(NW * 64), 0, st, a);
  else
    hipLaunchKernelGGL((mips_filter_f16<QW, NW, INCLUSIVE, false>), g, dim3(NW * 64), 0, st, a);
}

hipError_t launch_filter(const FilterArgs& a, int qw, bool inclusive, unsigned grid, hipStream_t st) {
  dim3 g(grid);
  if (a.compact) {
    if (qw != 2 || inclusive || a.ub) return hipErrorInvalidValue;
    hipLaunchKernelGGL((mips_filter_f16<2, 8, false, false, true>), g, dim3(8 * 64), 0, st, a);
    return hipGetLastError();
  }
  if (qw == 4) {
    if (inclusive)
      launch_filter_bounded<4, 4, true>(a, g, st);
    else
      launch_filter_bounded<4, 4, false>(a, g, st);
  } else if (qw == 2) {
    if (inclusive)
      launch_filter_bounded<2, 8, true>(a, g, st);
    else
      launch_filter_bounded<2, 8, false>(a, g, st);
  } else {
    if (inclusive)
      launch_filter_bounded<1, 8, true>(a, g, st);
    else
      launch_filter_bounded<1, 8, false>(a, g, st);
  }
  return hipGetLastError();
}

// deep: the one-workgroup-per-CU form with four pair buffers (QW = 1 only; the caller sizes the grid for it)
hipError_t launch_filter_i8(const FilterArgsI8& a, int qw, unsigned grid, hipStream_t st, bool deep) {
  if (qw == 2)   // (flag 256, experiment: 24 KiB of unused dynamic LDS leave room for ONE workgroup per CU)
    hipLaunchKernelGGL((mips_filter_i8<2, 2>), dim3(grid), dim3(kFilterThreads), (a.flags & 256u) ? 24576 : 0, st, a);
  else if (qw == 1 && a.q_blocks) {   // row-split: 1, 2 or 4 query blocks replicated over the eight waves
    if (a.q_blocks != 1 && a.q_blocks != 2 && a.q_blocks != 4) return hipErrorInvalidValue;
    if (deep)
      hipLaunchKernelGGL((mips_filter_i8<1, 4, true>), dim3(grid), dim3(kFilterThreads), 0, st, a);
    else
      hipLaunchKernelGGL((mips_filter_i8<1, 2, true>), dim3(grid), dim3(kFilterThreads), 0, st, a);
  } else if (qw == 1 && deep)
    hipLaunchKernelGGL((mips_filter_i8<1, 4>), dim3(grid), dim3(kFilterThreads), 0, st, a);
  else if (qw == 1)
    hipLaunchKernelGGL((mips_filter_i8<1, 2>), dim3(grid), dim3(kFilterThreads), 0, st, a);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_column_stats(const void* xb16, long long n, float* partial, float* col, QuantStats* stats, hipStream_t st) {
  hipLaunchKernelGGL(column_stats_partial, dim3(kColStatGroups), dim3(256), 0, st, (const _Float16*)xb16, n, partial, stats);
  hipLaunchKernelGGL(column_stats_finish, dim3(1), dim3(kDim), 0, st, partial, kColStatGroups, n, col, stats);
  return hipGetLastError();
}

hipError_t launch_quantise_rows_i8(const void* xb16, long long n, const float* col, signed char* xb8, float2* blk, QuantStats* stats,
                                   hipStream_t st) {
  if (n == 0) return hipSuccess;
  const long long want = ((n + 31) / 32 + 3) / 4;
  const unsigned grid = (unsigned)std::min<long long>(want, 8ll * 1024);
  hipLaunchKernelGGL(quantise_rows_i8, dim3(grid), dim3(256), 0, st, (const _Float16*)xb16, n, col, xb8, blk, stats);
  return hipGetLastError();
}

hipError_t launch_prep_queries_i8(const void* xq_pad16, long long nq_pad, const float* col, const QuantStats* stats,
                                  signed char* xq8, NominateParams* qp, unsigned long long* stat_nom, hipStream_t st) {
  if (nq_pad == 0) return hipSuccess;
  hipLaunchKernelGGL(prep_queries_i8, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, (const _Float16*)xq_pad16, nq_pad, col,
                     stats, xq8, qp, stat_nom);
  return hipGetLastError();
}

#ifdef PROQA_FILTER_STAMPS
void read_filter_stamps(unsigned long long* out5) {   // (eight words)
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_filter_stamps), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_filter_stamps), z, sizeof z);
}
#endif

hipError_t launch_advance_page(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int page_k,
                               unsigned long long* bound_keys, float* ub, unsigned char* done, const float* margin,
                               float* ub_filter, hipStream_t st) {
  hipLaunchKernelGGL(advance_page, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, run_keys, run_n, nq, page_k,
                     bound_keys, ub, done, margin, ub_filter);
  return hipGetLastError();
}

hipError_t launch_upconvert_f16_to_f32(const void* src, float* dst, long long n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const long long nthreads = (n + 7) / 8;
  hipLaunchKernelGGL(upconvert_rows_f16_to_f32, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                     (const _Float16*)src, dst, n);
  return hipGetLastError();
}

hipError_t launch_row_norm_stats(const float* xb32, const void* xb16, long long n_rows, unsigned* norm_stats,
                                 hipStream_t st) {
  if (n_rows == 0) return hipSuccess;
  hipLaunchKernelGGL(row_norm_stats, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, xb32,
                     (const _Float16*)xb16, n_rows, norm_stats);
  return hipGetLastError();
}

hipError_t launch_query_margins(const void* xq, int dtype, long long nq, long long nq_pad, const unsigned* norm_stats,
                                float* xq32, float* margin, const float* tau, float* tau_filter, hipStream_t st) {
  if (nq_pad == 0) return hipSuccess;
  hipLaunchKernelGGL(query_margins, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, xq, dtype, nq, nq_pad,
                     norm_stats, xq32, margin, tau, tau_filter);
  return hipGetLastError();
}

hipError_t launch_bootstrap(const char* xb, const void* xq_pad, int n_rows, unsigned nq_pad, int k, float* scores,
                            unsigned long long* run_keys, unsigned* run_n, float* tau, unsigned long long* stat,
                            unsigned* overflow, hipStream_t st, int run_stride, int tau_rank) {
  if (run_stride <= 0) run_stride = k;
  if (tau_rank <= 0 || tau_rank > k) tau_rank = k;
  const int s_stride = (n_rows + 31) / 32 * 32;
  const unsigned row_tiles = (unsigned)(s_stride / 32);
  const unsigned per_wg = kBootWaves * kBootTilesPerWave;
  hipLaunchKernelGGL(bootstrap_scores, dim3((row_tiles + per_wg - 1) / per_wg, nq_pad / 32), dim3(kBootWaves * 64), 0, st,
                     xb, xq_pad, n_rows, s_stride, scores);
#define PROQA_SELECT_CASE(E)                                                                                            \
  hipLaunchKernelGGL(bootstrap_select<E>, dim3(nq_pad), dim3(kMergeThreads), 0, st, scores, n_rows, s_stride, k, run_stride, \
                     run_keys, run_n, tau, stat, overflow, tau_rank)
  if (n_rows <= 4 * kMergeThreads)
    PROQA_SELECT_CASE(4);
  else if (n_rows <= 8 * kMergeThreads)
    PROQA_SELECT_CASE(8);
  else if (n_rows <= 16 * kMergeThreads)
    PROQA_SELECT_CASE(16);
  else
    PROQA_SELECT_CASE(32);
#undef PROQA_SELECT_CASE
  static_assert(kBootstrapMaxRows == 32 * kMergeThreads, "largest select");
  return hipGetLastError();
}

hipError_t launch_gather_index_rows(const void* xb16, const float* xb32, long long n_index, const long long* ids, long long n,
                                    long long idx_offset, void* out, bool out_f32, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const long long pieces = n * (out_f32 ? kDim / 4 : kDim / 8);
  hipLaunchKernelGGL(gather_index_rows, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const _Float16*)xb16, xb32,
                     n_index, ids, n, idx_offset, out, out_f32 ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_rescore_nominated_lists(const RescoreArgs& a, unsigned groups, unsigned nq, hipStream_t st) {
  if (groups == 0 || nq == 0) return hipSuccess;
  if (a.lists_per_group == 0 || a.lists_per_group > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rescore_nominated_lists, dim3(groups, nq), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_gather_query_rows(const void* xq_pad, const int* ids, int n, void* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_query_rows, dim3((unsigned)((n * (kRowBytes / 16) + 255) / 256)), dim3(256), 0, st, (const uint4*)xq_pad, ids, n,
                     (uint4*)out);
  return hipGetLastError();
}
hipError_t launch_scatter_result_rows(const float* D_src, const long long* I_src, const int* ids, int n, int k, float* D, long long* I,
                                      int out_stride, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_result_rows, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, st, D_src, I_src, ids, n, k, D, I, out_stride);
  return hipGetLastError();
}

hipError_t launch_flag_short_lists(const unsigned* run_n, long long nq, unsigned want, unsigned* flag, hipStream_t st) {
  hipLaunchKernelGGL(flag_short_lists, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, run_n, nq, want, flag);
  return hipGetLastError();
}

hipError_t launch_merge(const MergeArgs& a, unsigned nq_pad, hipStream_t st) {
  const bool big = a.sort_cap > kMaxSortKeys;
  if (a.xb16) {   // records of the int8 nomination scan
    if (big || a.xq32 || a.compact || a.inclusive || a.bound_keys) return hipErrorInvalidValue;
    // Which merge: (a) at most 256 queries -- the chip is empty during their merges, whose time is a chain of dependent round
    // trips: a 1024-thread workgroup per query re-scores a round's nominated rows in ONE trip of 16 waves instead of three
    // of four (PROQA_MERGE_NOM_WIDE=0: the 256-thread form, A/B); (b) rounds that nominate few rows (a.nom_keys == 1024: k x
    // growth <= 200, e.g. the k = 80 search of thousands of queries) -- the merge that holds 1024 keys, EIGHT workgroups per CU,
    // i.e. the 2032 merges of a round resident at once instead of 1792 + a second wave of 240 (PROQA_MERGE_NOM_CAP=2048: the
    // seven-per-CU form, A/B); (c) 2048 keys, seven per CU.  A round that nominates more rows than its merge holds is re-scanned
    // on the overflow-safe path AND suspends the int8 rounds: (b) is only taken with a factor ~4 of headroom.
    static const bool kSmallNom = !(getenv("PROQA_MERGE_NOM_CAP") && atoi(getenv("PROQA_MERGE_NOM_CAP")) == 2048);
    static const bool kWideNom = !(getenv("PROQA_MERGE_NOM_WIDE") && atoi(getenv("PROQA_MERGE_NOM_WIDE")) == 0);
    if (kWideNom && nq_pad <= 256)
      hipLaunchKernelGGL((topk_merge<kMergeNominatedI8, kMaxSortKeys, kOnePassMergeThreads>), dim3(nq_pad), dim3(kOnePassMergeThreads), 0,
                         st, a);
    else if (kSmallNom && a.nom_keys == kNominatedSortKeys)
      hipLaunchKernelGGL((topk_merge<kMergeNominatedI8, kNominatedSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
    else
      hipLaunchKernelGGL((topk_merge<kMergeNominatedI8, kMaxSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
  } else if (a.xq32) {
    if (big)
      hipLaunchKernelGGL((topk_merge<kMergeExactF32, kBigSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
    else
      hipLaunchKernelGGL((topk_merge<kMergeExactF32, kMaxSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
  } else {
    if (a.sort_cap > kBigSortKeys)
      hipLaunchKernelGGL((topk_merge<kMergeF16, kOnePassSortKeys, kOnePassMergeThreads>), dim3(nq_pad), dim3(kOnePassMergeThreads), 0, st, a);
    else if (a.sort_cap > kMidSortKeys)
      hipLaunchKernelGGL((topk_merge<kMergeF16, kBigSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
    else if (big)
      hipLaunchKernelGGL((topk_merge<kMergeF16, kMidSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
    else
      hipLaunchKernelGGL((topk_merge<kMergeF16, kMaxSortKeys>), dim3(nq_pad), dim3(kMergeThreads), 0, st, a);
  }
  return hipGetLastError();
}

// developer switch, read once: PROQA_DEBUG_NOHIT keeps every threshold at +inf so that the filter never
// logs a candidate (measures the pure scan)
static const bool kDebugNoHit = getenv("PROQA_DEBUG_NOHIT") != nullptr;

hipError_t launch_prep_queries(const void* xq, int dtype, long long nq, long long nq_pad, void* xq_pad,
                               float* tau, unsigned* run_n, unsigned long long* stat, const unsigned char* done,
                               bool reset_stat, unsigned* inexact, unsigned* overflow, hipStream_t st, unsigned* short_rounds) {
  const long long n = nq_pad * kDim;
  hipLaunchKernelGGL(prep_queries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xq, dtype, nq, nq_pad,
                     (_Float16*)xq_pad, tau, run_n, stat, done, reset_stat ? 1 : 0, inexact,
                     kDebugNoHit ? 1 : 0, overflow, short_rounds);
  return hipGetLastError();
}

hipError_t launch_finalize(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int page_k,
                           long long idx_offset, float* D, long long* I, int out_stride, int out_offset,
                           const unsigned* overflow, unsigned* status, SearchMirror* mirror, const unsigned long long* stat,
                           hipStream_t st, const unsigned long long* stat_nominated) {
  const long long n = nq * page_k;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(finalize_topk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, run_keys, run_n,
                     nq, page_k, idx_offset, D, I, out_stride, out_offset, overflow, status, mirror, stat, stat_nominated);
  return hipGetLastError();
}

hipError_t launch_merge_lists(const float* D_parts, const long long* I_parts, int n_parts, long long nq,
                              int k, long long stride_d, long long stride_i, float* D, long long* I, hipStream_t st,
                              const unsigned* status_src, long long status_stride, unsigned* status_host, bool parts_sorted) {
  if (nq == 0) {   // nothing to merge, but the caller still reads every part's status word
    if (status_host && status_src)
      hipLaunchKernelGGL(copy_status_words, dim3(1), dim3(256), 0, st, status_src, status_stride, n_parts, status_host);
    return hipGetLastError();
  }
  const long long per_q = (long long)n_parts * k;
  if (parts_sorted && per_q <= kRankMergeKeys && n_parts <= kRankMergeParts) {
    hipLaunchKernelGGL(merge_sorted_lists, dim3((unsigned)nq), dim3(kMergeThreads), (size_t)per_q * 8, st, D_parts, I_parts,
                       n_parts, nq, k, stride_d, stride_i, D, I, status_src, status_stride, status_host);
    return hipGetLastError();
  }
  if (per_q <= kMaxMergeListKeys) {
    unsigned P = kMergeThreads;
    while (P < (unsigned)per_q) P <<= 1;
    const size_t lds = (size_t)P * 8;
    hipError_t e = hipSuccess;
#define PROQA_MERGE_CASE(NK)                                                                                        \
  case NK:                                                                                                          \
    if (lds > 64 * 1024)                                                                                            \
      e = hipFuncSetAttribute((const void*)merge_lists<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
    if (e == hipSuccess)                                                                                            \
      hipLaunchKernelGGL(merge_lists<NK>, dim3((unsigned)nq), dim3(kMergeThreads), lds, st, D_parts, I_parts,       \
                         n_parts, nq, k, stride_d, stride_i, D, I, status_src, status_stride, status_host);               \
    break;
    switch (P / kMergeThreads) {
      PROQA_MERGE_CASE(1)
      PROQA_MERGE_CASE(2)
      PROQA_MERGE_CASE(4)
      PROQA_MERGE_CASE(8)
      PROQA_MERGE_CASE(16)
      PROQA_MERGE_CASE(32)
      PROQA_MERGE_CASE(64)
      default: return hipErrorInvalidValue;
    }
#undef PROQA_MERGE_CASE
    static_assert(kMaxMergeListKeys == 64 * kMergeThreads, "largest list merge");
    return e != hipSuccess ? e : hipGetLastError();
  }
  if (status_host)
    hipLaunchKernelGGL(copy_status_words, dim3(1), dim3(256), 0, st, status_src, status_stride, n_parts, status_host);
  // segmented radix sort in HBM, in query chunks of <= 2^27 keys (1 GiB per key buffer)
  const long long chunk_q = std::max<long long>(1, std::min<long long>(nq, (1ll << 27) / per_q));
  unsigned long long *keys_in = nullptr, *keys_out = nullptr;
  int* offsets = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipSuccess;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(st);
    if (keys_in) (void)hipFree(keys_in);
    if (keys_out) (void)hipFree(keys_out);
    if (offsets) (void)hipFree(offsets);
    if (tmp) (void)hipFree(tmp);
  };
#define MERGE_TRY(expr) if ((e = (expr)) != hipSuccess) { cleanup(); return e; }
  MERGE_TRY(hipMalloc((void**)&keys_in, (size_t)chunk_q * per_q * 8));
  MERGE_TRY(hipMalloc((void**)&keys_out, (size_t)chunk_q * per_q * 8));
  MERGE_TRY(hipMalloc((void**)&offsets, (size_t)(chunk_q + 1) * sizeof(int)));
  hipLaunchKernelGGL(segment_offsets, dim3((unsigned)((chunk_q + 256) / 256)), dim3(256), 0, st, chunk_q, (int)per_q,
                     offsets);
  MERGE_TRY(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, tmp_bytes, keys_in, keys_out,
                                                                  (int)(chunk_q * per_q), (int)chunk_q, offsets,
                                                                  offsets + 1, 0, 64, st));
  MERGE_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  for (long long q0 = 0; q0 < nq; q0 += chunk_q) {
    const long long m = std::min(chunk_q, nq - q0);
    const long long items = m * per_q;
    hipLaunchKernelGGL(pack_part_keys, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, D_parts, I_parts,
                       n_parts, nq, k, stride_d, stride_i, q0, m, keys_in);
    MERGE_TRY(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(tmp, tmp_bytes, keys_in, keys_out, (int)items,
                                                                    (int)m, offsets, offsets + 1, 0, 64, st));
    hipLaunchKernelGGL(emit_sorted_prefix, dim3((unsigned)((m * k + 255) / 256)), dim3(256), 0, st, keys_out, I_parts,
                       n_parts, nq, k, stride_d, stride_i, q0, m, D, I);
    MERGE_TRY(hipGetLastError());
  }
#undef MERGE_TRY
  cleanup();
  return hipSuccess;
}

hipError_t launch_convert_f32_to_f16(const float* src, void* dst, long long n, unsigned* inexact, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const long long nthreads = (n + 3) / 4;
  hipLaunchKernelGGL(convert_rows_f32_to_f16, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                     src, (_Float16*)dst, n, inexact);
  return hipGetLastError();
}

}  // namespace proqa

// Dense layer of the encoder as a hand-written CDNA4 GEMM with a fused epilogue:
//     Y[M,N] = epilogue( X[M,K] . W[N,K]^T + bias[N] ),   fp16 in / fp32 accumulate / fp16 out
// X = token rows, W = the [out,in] weight of the reference checkpoint (both K-contiguous: a "TN" product).
// Replaces the rocBLAS call + the separate bias_gelu pass of BertIntermediate
// (/root/reference/retrieval/retriever.py:41 -> transformers BertIntermediate: dense + erf GELU).
//
// Structure (gfx950, wave64):
//   * workgroup = 8 waves, tile 256 (n) x 256 (m), K-step 64; a wave owns 128 n x 64 m = 4 x 2 MFMA tiles of
//     v_mfma_f32_32x32x16_f16 (128 accumulator registers).  The WEIGHT rows are the MFMA A operand and the TOKEN
//     rows the B operand, so a lane owns one token row m and, per tile, 16 output features n: with the A rows of a
//     64-feature block taken in the order  n = 32*half + 16*tile + r,  the 32 accumulators a lane holds for that
//     block are 32 CONSECUTIVE features of its row -- the epilogue stores them as four 16-byte pieces, no LDS
//     transpose, no 2-byte stores.
//   * both operand tiles of a K-step (2 x 32 KiB) are fetched by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//     wave-instruction) into a two-deep ring; the 16-byte pieces of a 128-byte tile row are XOR-swizzled with
//     (row >> 1) & 7 on the DMA SOURCE address (the LDS image stays lane-linear), which makes every
//     ds_read_b128 fragment load conflict-free for both the permuted A rows and the plain B rows.
//   * the schedule of the K loop (two wave groups in anti-phase, region-wise ring refill, counted vmcnt) is described
//     in front of the kernel.
//   * persistent workgroups, one per CU; tile order is XCD-aware: block b runs on XCD b % 8 and walks the
//     256-row token slabs  b%8, b%8 + 8, ...  feature tile by feature tile, so a token slab is fetched from HBM
//     by ONE XCD's L2 and re-read from there by that XCD's workgroups; the (small) weight is shared by all.
//   * the K-steps of consecutive output tiles form ONE prefetch stream (two steps ahead), across the epilogues.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 256;            // output tile edge (features and token rows)
constexpr int kBK = 64;               // K-step
constexpr int kRowB = kBK * 2;        // 128 bytes of a tile row per K-step
constexpr int kOpBytes = kTile * kRowB;   // 32 KiB: one operand tile of one K-step
constexpr int kWaves = 8;

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2 };

// LDS row (within a 64-row block of the weight tile) that MFMA A-row i of 32-row tile j reads: see the header
__device__ __forceinline__ int a_row(int i, int j) { return 32 * ((i >> 2) & 1) + 16 * j + 4 * (i >> 3) + (i & 3); }

// ds_read_b128 / counted s_waitcnt as inline assembly: hipcc waits lgkmcnt(0) for a fragment even when six newer
// reads are in flight behind it, which exposes an LDS round trip in front of every group of MFMAs.  A read issued
// this way is invisible to the compiler's wait insertion; lds_wait<N>() names the fragments it retires, so every
// use of them is ordered after the wait.
template <int IMM>
__device__ __forceinline__ f16x8 lds_read128(unsigned addr) {
  f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait(f16x8 (&fa)[4], f16x8 (&fb)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1])
               : "n"(N)
               : "memory");
}

// erf-form GELU (hidden_act = 'gelu'):  gelu(x) = x Phi(x) = max(x, 0) - a Phi(-a),  a = |x|.
// Phi(-a) = exp2(q(a)) with q a degree-6 minimax fit of log2 Phi(-a) on [0, 6] (|error| <= 6.5e-5 in the exponent: a
// RELATIVE error of 4.6e-5 of Phi(-a), so the negative tail keeps its relative accuracy; the fp16 result differs from the
// correctly rounded one in 2 % of all fp16 inputs, by one ulp -- an erf formula with a small ABSOLUTE error such as
// Abramowitz & Stegun 7.1.26 loses the tail and needs a reciprocal on top).  Beyond a = 6 (Phi(-6) = 1e-9) a is clamped in
// both factors: the correction is below the fp16 resolution of x, and of 0.  9 plain VALU operations + one exp2 per
// element where libdevice's erff costs ~40; coefficients from scripts/dev_gelu_fit.py.
__device__ __forceinline__ float gelu_erf(float x) {
  const float a = __builtin_fminf(__builtin_fabsf(x), 6.0f);
  float q = __builtin_fmaf(2.299005791428499e-05f, a, -0.000611100229434669f);
  q = __builtin_fmaf(q, a, 0.007195569109171629f);
  q = __builtin_fmaf(q, a, -0.05118535831570625f);
  q = __builtin_fmaf(q, a, -0.46127188205718994f);
  q = __builtin_fmaf(q, a, -1.1501742601394653f);
  q = __builtin_fmaf(q, a, -1.000064730644226f);
  const float u = __builtin_amdgcn_exp2f(q);                   // Phi(-a)
  return __builtin_fmaf(-a, u, __builtin_fmaxf(x, 0.0f));
}

// The same on packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction at the scalar forms' rate): the seven
// fused multiply-adds and the bias add of an element pair become eight instructions instead of sixteen (~1200 instead of
// ~1600 VALU instructions per wave and 256 x 256 tile).  Same operations in the same order per element: bit-identical to
// gelu_erf.
typedef float f32x4e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4e splat4(float v) { return f32x4e{v, v, v, v}; }
// FOUR elements per call: every Horner step is applied to all four before the next one, i.e. two independent v_pk_fma_f32
// per step (eight at a time spill: the kernel sits at 252 VGPRs) -- written element pair by element pair, hipcc ran each pair's seven dependent
// instructions through one register pair back to back (the kernel sits at 252 VGPRs) and the epilogue stayed latency-bound:
// packed arithmetic alone measured -1 % (ABLATIONS R6.7).
// Written in an = -a = max(-|x|, -6) (one v_max_f32 with source modifiers): the Horner steps of q alternate in sign -- every
// odd intermediate is the exact negative of gelu_erf's, fused multiply-adds round symmetrically -- and the last step is
// fma(an, u, m) without a negation (hipcc turns a negated four-vector into 128 v_xor_b32 per tile).
__device__ __forceinline__ f32x4e gelu_erf4(f32x4e x) {
  f32x4e an;
#pragma unroll
  for (int e = 0; e < 4; ++e) an[e] = __builtin_fmaxf(-__builtin_fabsf(x[e]), -6.0f);
  f32x4e q = __builtin_elementwise_fma(splat4(2.299005791428499e-05f), an, splat4(0.000611100229434669f));   // -t1
  q = __builtin_elementwise_fma(q, an, splat4(0.007195569109171629f));                                        //  t2
  q = __builtin_elementwise_fma(q, an, splat4(0.05118535831570625f));                                         // -t3
  q = __builtin_elementwise_fma(q, an, splat4(-0.46127188205718994f));                                        //  t4
  q = __builtin_elementwise_fma(q, an, splat4(1.1501742601394653f));                                          // -t5
  q = __builtin_elementwise_fma(q, an, splat4(-1.000064730644226f));                                          //  q(a)
  f32x4e u, m;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    u[e] = __builtin_amdgcn_exp2f(q[e]);   // Phi(-a)
    m[e] = __builtin_fmaxf(x[e], 0.0f);
  }
  return __builtin_elementwise_fma(an, u, m);
}

// The main loop runs the two waves of every SIMD in ANTI-PHASE ("ping-pong").  (Rounds 1-3 had all eight waves read their
// fragments at the same time and then all issue MFMAs at the same time, one barrier per K-step: the matrix pipe of a SIMD
// idled while its two waves waited for LDS; bit-identical results, 3-4 % slower.)  Waves 0-3 (group X, one per SIMD) and waves 4-7
// (group Y, their SIMD partners) run the same instruction stream ONE INTERVAL apart: every interval ends with a workgroup
// barrier, a wave alternates LOAD intervals (ds_reads of the next quadrant + its share of the LDS-DMA prefetch) with
// COMPUTE intervals (8 MFMAs = one 64 x 32 quadrant of its 128 x 64 tile over the whole K-step), and while X computes Y
// loads.  Per K-step a wave runs 4 phases (quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0)); its LOAD intervals read 8 / 4 / 8 / 4
// fragments (A0 | B1 | A1 | B0 of the NEXT step).
//   Ring discipline (two 64 KiB K-step buffers, prefetch distance TWO steps): a buffer is refilled region by region as
// soon as BOTH groups have read that region -- of step t+2, the B0 rows (the first 32 of every 64-row token quarter; their
// last reader was phase 3 of step t-1) are requested in phase 0 of step t, the A0 rows in phase 1, the B1 rows in phase 2,
// the A1 rows in phase 3: two 1 KiB pieces per wave and LOAD interval -- so a DMA has 8+ intervals (~2000 cycles) to land.
// A wave waits for its DMAs of step t+1 with a COUNTED vmcnt in phase 2 of step t (vmcnt(4): the four pieces of step t+2
// issued in phases 0 and 1 stay in flight; never 0 in the loop; the 16 stores of an epilogue are counted too), the reads of
// that data start in phase 3, one barrier later for X and two for Y.  Barriers are raw s_barrier (no vmcnt drain).
//   At the end of an output tile X takes one extra barrier (the groups fall in step), both run the epilogue at the same
// time, then Y takes one extra barrier (anti-phase again).
template <int EPI, int DBG = 0, bool PK = true>   // PK: packed-fp32 epilogue arithmetic (false: the scalar form, developer A/B).  DBG (timing experiments, wrong results): bit 0 = no fragment reads after the first step, bit 1 = no DMA after the prologue, bit 2 = no epilogue, bit 3 = every DMA from the same 64 KiB
__global__ __launch_bounds__(kWaves * 64) void gemm_tn_f16(const _Float16* __restrict__ X, const _Float16* __restrict__ W,
                                                             const _Float16* __restrict__ bias, _Float16* __restrict__ Y,
                                                             int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * kOpBytes + kWaves * 4096];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int wn = wave >> 2;   // which 128-feature half of the tile; also the phase group (0 = X, 1 = Y)
  const int wm = wave & 3;    // which 64-row quarter of the tile

  const int tiles_n = N / kTile, tiles_m = M / kTile;
  const int n_steps = K / kBK;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, n_slots = (gridDim.x + 7 - xcd) >> 3;
  const int my_slabs = (tiles_m - xcd + 7) >> 3;
  const int my_tiles = my_slabs * tiles_n;   // (<= 2^22 slabs x a few feature tiles)

  // ---- LDS-DMA pieces of this wave (1 KiB = 8 tile rows each; source-side XOR swizzle as above)
  //   A0 region (rows 0-63 and 128-191): pieces 2 wave, 2 wave + 1 of its 16;  A1 region: the same rows + 64
  //   B0 region (rows 64 g + [0, 32), g = 0..3): pieces 2 wave, 2 wave + 1 of its 16;  B1 region: the same rows + 32
  const int dma_r = lane >> 3, dma_p = lane & 7;
  int a_src[2], b_src[2], a_lds[2], b_lds[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int q = 2 * wave + e;
    const int row0 = q < 8 ? 8 * q : 128 + 8 * (q - 8);
    const int row = row0 + dma_r;
    a_src[e] = row * K * 2 + ((dma_p ^ ((row >> 1) & 7)) << 4);
    a_lds[e] = row0 * kRowB;
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int q = 2 * wave + e;
    const int row0 = 64 * (q >> 2) + 8 * (q & 3);
    const int row = row0 + dma_r;
    b_src[e] = row * K * 2 + ((dma_p ^ ((row >> 1) & 7)) << 4);      // (+32 rows: the same swizzle term)
    b_lds[e] = kOpBytes + row0 * kRowB;
  }
  auto dma = [&](const char* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + lds_off), 16, 0, 0);
  };
  // region 0: A0 rows, 1: B0 rows, 2: A1 rows, 3: B1 rows of K-step `step` of the tile at (wbase, xbase) into ring buffer `buf`
  auto issue_region = [&](int region, const char* wbase, const char* xbase, int step, int buf) {
    const long long koff = (DBG & 8) ? 0 : (long long)step * kRowB;   // DBG 8: every DMA re-reads the same 64 KiB (L2 hits)
    if (DBG & 8) { wbase = (const char*)W; xbase = (const char*)X; }
    const int bo = buf * 2 * kOpBytes;
    if (region == 0) {
#pragma unroll
      for (int e = 0; e < 2; ++e) dma(wbase + koff + a_src[e], bo + a_lds[e]);
    } else if (region == 2) {
#pragma unroll
      for (int e = 0; e < 2; ++e) dma(wbase + koff + a_src[e] + 64ll * K * 2, bo + a_lds[e] + 64 * kRowB);
    } else if (region == 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) dma(xbase + koff + b_src[e], bo + b_lds[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) dma(xbase + koff + b_src[e] + 32ll * K * 2, bo + b_lds[e] + 32 * kRowB);
    }
  };

  // ---- fragment read addresses (see gemm_tn_f16): A tiles of block 0 at rows +0 / +16, of block 1 at +64 / +80
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  unsigned a_addr[4], b_addr[4];
  {
    const int arow = wn * 128 + a_row(li, 0), brow = wm * 64 + li;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a_addr[s] = lds0 + arow * kRowB + (((2 * s + half) ^ ((arow >> 1) & 7)) << 4);
      b_addr[s] = lds0 + kOpBytes + brow * kRowB + (((2 * s + half) ^ ((brow >> 1) & 7)) << 4);
    }
  }
  // fa[j][s]: A tile j of the current 64-row block, k16 sub-step s; fb[s]: the current 32-row B tile
  auto read_a = [&](f16x8 (&fa)[2][4], int blk, int buf) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned aa = a_addr[s] + buf * 2 * kOpBytes;
      if (blk == 0) {
        fa[0][s] = lds_read128<0>(aa);
        fa[1][s] = lds_read128<16 * kRowB>(aa);
      } else {
        fa[0][s] = lds_read128<64 * kRowB>(aa);
        fa[1][s] = lds_read128<80 * kRowB>(aa);
      }
    }
  };
  auto read_b = [&](f16x8 (&fb)[4], int mt, int buf) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned bb = b_addr[s] + buf * 2 * kOpBytes;
      fb[s] = mt == 0 ? lds_read128<0>(bb) : lds_read128<32 * kRowB>(bb);
    }
  };

  // (groups of 3, 4, 9 or 12 feature tiles measure the same on all four encoder shapes)
  constexpr int kGroup = 8;
  auto tile_bases = [&](int t, const char*& wbase, const char*& xbase, int& m0, int& n0) {
    const unsigned per_full_group = (unsigned)my_slabs * kGroup;
    const int grp = (int)((unsigned)t / per_full_group);
    const int g = tiles_n - grp * kGroup < kGroup ? tiles_n - grp * kGroup : kGroup;
    const unsigned r = (unsigned)t - grp * per_full_group;
    const int slab = xcd + 8 * (int)(r / (unsigned)g);
    const int nt = grp * kGroup + (int)(r % (unsigned)g);
    m0 = slab * kTile;
    n0 = nt * kTile;
    wbase = (const char*)W + (long long)n0 * K * 2;
    xbase = (const char*)X + (long long)m0 * K * 2;
  };

  int t = slot;
  if (t >= my_tiles) return;
  const char *wbase, *xbase;
  int m0, n0;
  tile_bases(t, wbase, xbase, m0, n0);

  // prefetch cursor: the K-step whose DMA is issued next (two steps ahead of the one being computed)
  int pf_t = t;
  int pf_step = 0, pf_buf = 0;
  const char *pf_w = wbase, *pf_x = xbase;
  bool pf_valid = true;
  auto pf_advance = [&]() {
    pf_buf ^= 1;
    if (++pf_step == n_steps) {
      pf_step = 0;
      pf_t += n_slots;
      pf_valid = pf_t < my_tiles;
      if (pf_valid) {
        int um, un;
        tile_bases(pf_t, pf_w, pf_x, um, un);
      }
    }
  };
  // prologue: steps 0 and 1 whole, then the first B0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (pf_valid) {
      issue_region(0, pf_w, pf_x, pf_step, pf_buf);
      issue_region(1, pf_w, pf_x, pf_step, pf_buf);
      issue_region(2, pf_w, pf_x, pf_step, pf_buf);
      issue_region(3, pf_w, pf_x, pf_step, pf_buf);
      pf_advance();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f16x8 fa[2][4], fb0[4], fb1[4], fb0n[4];
  read_b(fb0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fb0[2]), "+v"(fb0[3])::"memory");
  int buf = 0;                 // ring buffer of the step being computed
  bool after_epilogue = false; // the epilogue's 16 stores sit between this step's waits and the DMAs they retire

  // interval ends
#define PP_END_LOAD()                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_barrier();                         \
  __builtin_amdgcn_sched_barrier(0)
#define PP_END_COMPUTE()                                \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_barrier();                         \
  __builtin_amdgcn_sched_barrier(0)

  for (; t < my_tiles; t += n_slots) {
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    if (wn == 1) __builtin_amdgcn_s_barrier();   // Y falls one interval behind X

    for (int step = 0; step < n_steps; ++step) {
      const bool last_of_all = step + 1 == n_steps && t + n_slots >= my_tiles;
      auto quadrant = [&](f16x8 (&a)[2][4], f16x8 (&b)[4], int blk, int mt) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[2 * blk][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][s], b[s], acc[2 * blk][mt], 0, 0, 0);
          acc[2 * blk + 1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][s], b[s], acc[2 * blk + 1][mt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
      };
      // ---- phase 0: A0, DMA B0 rows of step + 2 (their last reader was phase 3 of the previous step) -> (A0, B0)
      if (!(DBG & 1) || step == 0) read_a(fa, 0, buf);
      if (pf_valid && !(DBG & 2)) issue_region(1, pf_w, pf_x, pf_step, pf_buf);
      PP_END_LOAD();
      quadrant(fa, fb0, 0, 0);
      PP_END_COMPUTE();
      // ---- phase 1: B1, DMA A0 rows of step + 2 -> (A0, B1)
      if (!(DBG & 1) || step == 0) read_b(fb1, 1, buf);
      if (pf_valid && !(DBG & 2)) issue_region(0, pf_w, pf_x, pf_step, pf_buf);
      PP_END_LOAD();
      quadrant(fa, fb1, 0, 1);
      PP_END_COMPUTE();
      // ---- phase 2: A1, retire the DMAs of step + 1 (4 younger ones stay in flight), DMA B1 rows of step + 2 -> (A1, B1)
      if (!(DBG & 1) || step == 0) read_a(fa, 1, buf);
      if (!pf_valid)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (after_epilogue)
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (pf_valid && !(DBG & 2)) issue_region(3, pf_w, pf_x, pf_step, pf_buf);
      PP_END_LOAD();
      quadrant(fa, fb1, 1, 1);
      PP_END_COMPUTE();
      // ---- phase 3: B0 of the next step, DMA A1 rows of step + 2 -> (A1, B0)
      if (!last_of_all && (!(DBG & 1) || step == 0)) read_b(fb0n, 0, buf ^ 1);
      if (pf_valid) {
        if (!(DBG & 2)) issue_region(2, pf_w, pf_x, pf_step, pf_buf);
        pf_advance();
      }
      PP_END_LOAD();
      quadrant(fa, fb0, 1, 0);
      PP_END_COMPUTE();
#pragma unroll
      for (int s = 0; s < 4; ++s) fb0[s] = fb0n[s];
      buf ^= 1;
      after_epilogue = false;
    }
    if (wn == 0) __builtin_amdgcn_s_barrier();   // X waits one interval: both groups run the epilogue together

    const int cm0 = m0, cn0 = n0;
    const int tn = t + n_slots;
    if (tn < my_tiles) tile_bases(tn, wbase, xbase, m0, n0);

    // ---- epilogue (as in gemm_tn_f16)
    char* stage = lds + 2 * 2 * kOpBytes + wave * 4096;
    const int wr_sw = (li >> 1) & 7;
    const int rd_row = lane >> 3, rd_q = lane & 7;
    if (DBG & 4) {   // keep the accumulators alive with one store
      float z = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) z += acc[a][b][e];
      if (z == 123.456f) Y[tid] = (_Float16)z;
    } else {
    // the lane's 2 x 32 bias values, requested before anything is stored: a load issued between the stores of the two
    // 64-feature blocks would wait for the first block's stores (vmcnt retires in order)
    f16x8 bias8[2][4];
    if (EPI != EPI_NONE) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int c = 0; c < 4; ++c) bias8[blk][c] = *(const f16x8*)(bias + cn0 + wn * 128 + blk * 64 + 32 * half + 8 * c);
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int nb = cn0 + wn * 128 + blk * 64;
      float bv[32];
      if (EPI != EPI_NONE) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[8 * c + e] = (float)bias8[blk][c][e];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x16 v = acc[blk * 2 + j][mt];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            f16x8 o;
            if constexpr (PK) {   // packed fp32, four elements side by side
#pragma unroll
              for (int e0 = 0; e0 < 8; e0 += 4) {
                f32x4e x, b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  x[e] = v[8 * c + e0 + e];
                  b4[e] = EPI != EPI_NONE ? bv[16 * j + 8 * c + e0 + e] : 0.f;
                }
                if (EPI != EPI_NONE) x += b4;
                if (EPI == EPI_BIAS_GELU) x = gelu_erf4(x);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e0 + e] = (_Float16)x[e];
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float x = v[8 * c + e];
                if (EPI != EPI_NONE) x += bv[16 * j + 8 * c + e];
                if (EPI == EPI_BIAS_GELU) x = gelu_erf(x);
                o[e] = (_Float16)x;
              }
            }
            const int piece = 4 * half + 2 * j + c;
            *(f16x8*)(stage + li * 128 + ((piece ^ wr_sw) << 4)) = o;
          }
        }
        _Float16* dst = Y + (long long)(cm0 + wm * 64 + mt * 32) * N + nb + rd_q * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + rd_row;
          const f16x8 o = *(const f16x8*)(stage + row * 128 + ((rd_q ^ ((row >> 1) & 7)) << 4));
          __builtin_nontemporal_store(o, (f16x8*)(dst + (long long)row * N));
        }
      }
    }
    }
    after_epilogue = true;
  }
#undef PP_END_LOAD
#undef PP_END_COMPUTE
}

}  // namespace
}  // namespace proqa

using namespace proqa;

extern "C" {

int proqa_gemm_tn_f16(const void* x, const void* w, const void* bias, void* y, int64_t m, int n, int k, int epilogue,
                      void* stream) {
  if (!x || !w || !y) return fail(PROQA_EINVAL, "gemm_tn: NULL argument");
  if (epilogue < 0 || epilogue > 2 || (epilogue != 0 && !bias)) return fail(PROQA_EINVAL, "gemm_tn: bad epilogue %d", epilogue);
  if (m < 0 || n <= 0 || k <= 0 || m % kTile || n % kTile || k % kBK || m > (1ll << 30))
    return fail(PROQA_EINVAL, "gemm_tn: m=%lld n=%d k=%d must be multiples of %d / %d / %d", (long long)m, n, k, kTile, kTile,
                kBK);
  if (m == 0) return PROQA_OK;
  const long long tiles = (m / kTile) * (long long)(n / kTile);
  const unsigned grid = (unsigned)std::min<long long>(tiles, device_cu_count());
  hipStream_t st = as_stream(stream);
  const dim3 g(grid), b(kWaves * 64);
  // developer switch: cut experiments on the main loop (wrong results by design; scripts/dev_gemm_ablate.py)
  static const int kDbg = getenv("PROQA_GEMM_DBG") ? atoi(getenv("PROQA_GEMM_DBG")) : 0;
  if (kDbg) {
#define PP_DBG_CASE(D) case D: hipLaunchKernelGGL((gemm_tn_f16<EPI_NONE, D>), g, b, 0, st, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (_Float16*)y, (int)m, n, k); break;
    switch (kDbg) { PP_DBG_CASE(4) PP_DBG_CASE(5) PP_DBG_CASE(6) PP_DBG_CASE(7) PP_DBG_CASE(12) default: return fail(PROQA_EINVAL, "gemm_tn: PROQA_GEMM_DBG=%d is not built", kDbg); }
#undef PP_DBG_CASE
    PROQA_LAUNCH_CHECK();
    return PROQA_OK;
  }
  switch (epilogue) {
    case EPI_NONE:
      hipLaunchKernelGGL(gemm_tn_f16<EPI_NONE>, g, b, 0, st, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias,
                         (_Float16*)y, (int)m, n, k);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL(gemm_tn_f16<EPI_BIAS>, g, b, 0, st, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias,
                         (_Float16*)y, (int)m, n, k);
      break;
    default: {
      // developer A/B switch: PROQA_GEMM_EPI_SCALAR=1 runs the scalar-fp32 epilogue of rounds 2-5 (same results)
      static const bool kScalarEpi = getenv("PROQA_GEMM_EPI_SCALAR") && atoi(getenv("PROQA_GEMM_EPI_SCALAR")) != 0;
      if (kScalarEpi)
        hipLaunchKernelGGL((gemm_tn_f16<EPI_BIAS_GELU, 0, false>), g, b, 0, st, (const _Float16*)x, (const _Float16*)w,
                           (const _Float16*)bias, (_Float16*)y, (int)m, n, k);
      else
        hipLaunchKernelGGL(gemm_tn_f16<EPI_BIAS_GELU>, g, b, 0, st, (const _Float16*)x, (const _Float16*)w,
                           (const _Float16*)bias, (_Float16*)y, (int)m, n, k);
    }
  }
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

}  // extern "C"

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
//   * one barrier per K-step: it publishes the step that was prefetched during the previous one (explicit
//     vmcnt(0): LDS-DMA completion is tracked by vmcnt only) and frees the other buffer for the next prefetch.
//   * persistent workgroups, one per CU; tile order is XCD-aware: block b runs on XCD b % 8 and walks the
//     256-row token slabs  b%8, b%8 + 8, ...  feature tile by feature tile, so a token slab is fetched from HBM
//     by ONE XCD's L2 and re-read from there by that XCD's workgroups; the (small) weight is shared by all.
//   * the first K-step of the next tile is prefetched before the epilogue of the current one.
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

// erf-form GELU (hidden_act = 'gelu'), 0.5 x (1 + erf(x / sqrt 2)), with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, three orders of magnitude below the fp16 resolution of the result) on the hardware rcp / exp2:
// 13 VALU operations per element where libdevice's erff costs ~40 -- the epilogue of a 256 x 256 tile evaluates
// 128 of them per lane with the matrix pipe idle.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
  const float erf_abs = __builtin_fmaf(-p, e, 1.0f);            // erf(|x| / sqrt 2)
  return 0.5f * __builtin_fmaf(__builtin_fabsf(x), erf_abs, x);  // x sign(x) erf(|x|/sqrt 2) = |x| erf_abs
}

template <int EPI>
__global__ __launch_bounds__(kWaves * 64) void gemm_tn_f16(const _Float16* __restrict__ X, const _Float16* __restrict__ W,
                                                          const _Float16* __restrict__ bias, _Float16* __restrict__ Y,
                                                          int M, int N, int K) {
  // [buffer][operand]: operand 0 = weight tile (A), 1 = token tile (B); the only LDS object of the kernel
  // ... plus 4 KiB per wave of epilogue staging (160 KiB in all)
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * kOpBytes + kWaves * 4096];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int wn = wave >> 2;   // which 128-feature half of the tile
  const int wm = wave & 3;    // which 64-row quarter of the tile

  const int tiles_n = N / kTile, tiles_m = M / kTile;
  const int n_steps = K / kBK;
  // XCD-aware persistent schedule
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, n_slots = (gridDim.x + 7 - xcd) >> 3;
  const int my_slabs = (tiles_m - xcd + 7) >> 3;        // token slabs xcd, xcd + 8, ...
  const long long my_tiles = (long long)my_slabs * tiles_n;

  // ---- LDS-DMA addressing: wave w fetches rows [32 w, 32 w + 32) of both operand tiles, 8 rows (1 KiB) per instruction
  const int dma_r = lane >> 3;   // row within the 8-row piece
  const int dma_p = lane & 7;    // 16-byte slot within the 128-byte row
  int dma_src_off[4];            // byte offset of this lane's source piece relative to (tile row 0, k0)
  int dma_row[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = wave * 32 + e * 8 + dma_r;
    dma_row[e] = row;
    dma_src_off[e] = (dma_p ^ ((row >> 1) & 7)) * 16;
  }
  auto issue_step = [&](const char* wbase, const char* xbase, int step, int buf) {
    // wbase / xbase: first row of the tile, K offset 0; row pitch K*2 bytes
    const long long koff = (long long)step * kRowB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const char* src = wbase + (long long)dma_row[e] * K * 2 + koff + dma_src_off[e];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + (buf * 2 + 0) * kOpBytes + (wave * 4 + e) * 1024),
                                       16, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const char* src = xbase + (long long)dma_row[e] * K * 2 + koff + dma_src_off[e];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + (buf * 2 + 1) * kOpBytes + (wave * 4 + e) * 1024),
                                       16, 0, 0);
    }
  };

  // ---- fragment read addresses (LDS bytes relative to buffer 0 / weight tile), per k16 sub-step s: piece 2 s + half of
  // the lane's row, swizzled.  The wave's other tiles are whole multiples of 16 rows away (same swizzle term), i.e.
  // compile-time offsets: weight tiles at rows +0, +16, +64, +80 of the wave's 128, token tiles at rows +0, +32.
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
  auto read_frags = [&](f16x8 (&fa)[4], f16x8 (&fb)[2], int s, int buf) {
    const unsigned aa = a_addr[s] + buf * 2 * kOpBytes, bb = b_addr[s] + buf * 2 * kOpBytes;
    fa[0] = lds_read128<0>(aa);
    fa[1] = lds_read128<16 * kRowB>(aa);
    fa[2] = lds_read128<64 * kRowB>(aa);
    fa[3] = lds_read128<80 * kRowB>(aa);
    fb[0] = lds_read128<0>(bb);
    fb[1] = lds_read128<32 * kRowB>(bb);
  };

  // Tile order of this XCD's workgroups: the feature tiles are taken in groups of up to 8 (<= 3 MiB of weight, which
  // then stays in the XCD's 4 MiB L2 while the token slabs stream past); inside a group the tile index runs
  // feature-tile-fastest over (slab, feature tile), so the ~32 workgroups of the XCD that run side by side share
  // 8 weight tiles and 4 token slabs per K-step.  Index t enumerates (group, slab, feature tile in group).
  // (groups of 6 or 12 measure the same)
  constexpr int kGroup = 8;
  auto tile_bases = [&](long long t, const char*& wbase, const char*& xbase, int& m0, int& n0) {
    const long long per_full_group = (long long)my_slabs * kGroup;
    const int grp = (int)(t / per_full_group);                 // all groups but the last hold kGroup feature tiles
    const int g = tiles_n - grp * kGroup < kGroup ? tiles_n - grp * kGroup : kGroup;
    const long long r = t - grp * per_full_group;
    const int slab = xcd + 8 * (int)(r / g);
    const int nt = grp * kGroup + (int)(r % g);
    m0 = slab * kTile;
    n0 = nt * kTile;
    wbase = (const char*)W + (long long)n0 * K * 2;
    xbase = (const char*)X + (long long)m0 * K * 2;
  };

  long long t = slot;
  if (t >= my_tiles) return;
  const char *wbase, *xbase;
  int m0, n0;
  tile_bases(t, wbase, xbase, m0, n0);
  issue_step(wbase, xbase, 0, 0);

  for (; t < my_tiles; t += n_slots) {
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};

    for (int step = 0; step < n_steps; ++step) {
      const int buf = step & 1;
      dma_wait_barrier();   // step `step` has landed everywhere; every wave is done with the other buffer
      // Two fragment sets in ping-pong.  Order: reads(0) | DMA of the next step | reads(1), wait(0), MFMAs(0) |
      // reads(2), wait(1), MFMAs(1) | reads(3), wait(2), MFMAs(2) | wait(3), MFMAs(3): a sub-step's reads are a whole
      // sub-step (8 MFMAs) ahead of their use, and the waits are counted (6 newer reads may still be in flight).
      f16x8 fa0[4], fb0[2], fa1[4], fb1[2];
      read_frags(fa0, fb0, 0, buf);
      __builtin_amdgcn_sched_barrier(0);
      if (step + 1 < n_steps) issue_step(wbase, xbase, step + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      auto mfmas = [&](f16x8 (&fa)[4], f16x8 (&fb)[2]) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      };
      read_frags(fa1, fb1, 1, buf);
      lds_wait<6>(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(fa0, fb0, 2, buf);
      lds_wait<6>(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(fa1, fb1, 3, buf);
      lds_wait<6>(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      lds_wait<0>(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
    }

    // every wave is past its last LDS read of this tile before buffer 0 is refilled for the next one
    const int cm0 = m0, cn0 = n0;
    const long long tn = t + n_slots;
    __syncthreads();
    if (tn < my_tiles) {
      tile_bases(tn, wbase, xbase, m0, n0);
      issue_step(wbase, xbase, 0, 0);
    }

    // ---- epilogue: lane (li, half) owns token row m and, per 64-feature block, features 32 half .. 32 half + 31.
    // Stored straight from the registers a wave-instruction would write 64 scattered 16-byte pieces (32 rows x 2);
    // instead every 32-row x 64-feature sub-block (4 KiB of fp16) goes through a wave-private LDS buffer and leaves as
    // whole 128-byte row segments: 8 lanes per row, 8 rows per store instruction.
    char* stage = lds + 2 * 2 * kOpBytes + wave * 4096;
    const int wr_sw = (li >> 1) & 7;                  // swizzle of the row this lane writes (row = li)
    const int rd_row = lane >> 3, rd_q = lane & 7;    // row within an 8-row group / 16-byte piece this lane stores
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int nb = cn0 + wn * 128 + blk * 64;       // first feature of the block
      float bv[32];
      if (EPI != EPI_NONE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f16x8 b8 = *(const f16x8*)(bias + nb + 32 * half + 8 * c);
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[8 * c + e] = (float)b8[e];
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {       // tile j of the block holds features 16 j .. 16 j + 15 of the lane's 32
          const f32x16 v = acc[blk * 2 + j][mt];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = v[8 * c + e];
              if (EPI != EPI_NONE) x += bv[16 * j + 8 * c + e];
              if (EPI == EPI_BIAS_GELU) x = gelu_erf(x);
              o[e] = (_Float16)x;
            }
            const int piece = 4 * half + 2 * j + c;   // 16-byte piece of the 128-byte staged row
            *(f16x8*)(stage + li * 128 + ((piece ^ wr_sw) << 4)) = o;
          }
        }
        // the wave's LDS operations execute in program order: the reads below see every lane's writes
        _Float16* dst = Y + (long long)(cm0 + wm * 64 + mt * 32) * N + nb + rd_q * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + rd_row;
          const f16x8 o = *(const f16x8*)(stage + row * 128 + ((rd_q ^ ((row >> 1) & 7)) << 4));
          // non-temporal: 403 MB of output per BertIntermediate launch would otherwise push the weight group and the
          // token slabs out of the XCD's L2
          __builtin_nontemporal_store(o, (f16x8*)(dst + (long long)row * N));
        }
      }
    }
  }
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
  switch (epilogue) {
    case EPI_NONE:
      hipLaunchKernelGGL(gemm_tn_f16<EPI_NONE>, g, b, 0, st, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias,
                         (_Float16*)y, (int)m, n, k);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL(gemm_tn_f16<EPI_BIAS>, g, b, 0, st, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias,
                         (_Float16*)y, (int)m, n, k);
      break;
    default:
      hipLaunchKernelGGL(gemm_tn_f16<EPI_BIAS_GELU>, g, b, 0, st, (const _Float16*)x, (const _Float16*)w,
                         (const _Float16*)bias, (_Float16*)y, (int)m, n, k);
  }
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

}  // extern "C"

// Fused BERT self-attention for one layer (proqa_attention_f16 in proqa_hip.h):
//   ctx[b, q, h*64:(h+1)*64] = softmax(Q K^T / 8 + key_mask) V      (head_dim 64)
// Replaces BertSelfAttention inside BertForRetriever.get_embed
// (/root/reference/retrieval/retriever.py:37,41 -> transformers BertModel); the additive key
// mask comes from the right-padding of em_collate (retrieval/datasets.py:29-45,298-305).
//
// One 256-thread workgroup per (sequence, head, 128 queries): the K and V rows of that head are staged in LDS
// 128 keys at a time with full-line coalesced loads (V transposed on the way in); each wave owns one 32-query block.
// Both products run on v_mfma_f32_32x32x16_f16 in the "swapped" orientation so that every lane
// owns ONE query column of the accumulator:
//   S^T = K Q^T   : lane (q = lane&31) holds 16 keys of its query per 32-key tile, so the row max
//                   / row sum of the online softmax are lane-local (+ one xor-32 shuffle);
//   O^T = V^T P^T : P^T is consumed straight from those registers as the MFMA B operand, and the
//                   rescale by exp(m_old - m_new) and the final 1/l are per-lane scalars.
// The k-order of an MFMA step is free as long as both operands agree, so V^T fragments gather
// exactly the keys a lane's P registers hold (rows {0-3, 8-11} + 4*half + 16*step of the tile).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attention.h"
#include "common.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHeadDim = 64;
constexpr int kKStride = kHeadDim + 8;  // fp16 elements per K row in LDS (144 B: conflict-free b128)
// V is staged TRANSPOSED (V^T[d][key]) so that the four consecutive keys a lane needs are one
// 8-byte read; rows are s_pad + 4 keys long: (s_pad*2 + 8) bytes = 2 banks more per d row, which
// spreads the 32 d rows of a ds_read_b64 over distinct bank pairs
constexpr int kVtPad = 4;
// per-wave output staging tile [32 queries][64 + 8]: the lane-owned 8-byte pieces of O^T are
// transposed through LDS so that the context rows leave as full 128-byte lines
constexpr int kOutStride = kHeadDim + 8;

// Token layout: padded ([B, seq_len] rows, cu_seqlens == nullptr, seq_lens = valid keys per sequence) or
// packed (cu_seqlens[b] = first token row of sequence b, no padding rows exist; seq_len = longest sequence).
// qkv_bias (nullable): the [3*hidden] bias of the fused Q|K|V projection, added here when the GEMM that
// produced qkv had no bias epilogue (proqa_encoder_forward: rocBLAS).  Packed fp16 adds: the sum of two
// fp16 values rounded once to fp16 is exactly what an fp32 add followed by a conversion gives.
__device__ __forceinline__ f16x8 add_bias8(f16x8 v, const _Float16* __restrict__ bias) {
  if (bias) v = v + *(const f16x8*)bias;
  return v;
}

// A workgroup owns up to 128 QUERIES of one (sequence, head) -- one 32-query block per wave -- and streams the keys of the
// sequence through LDS in chunks of 128 (K rows + V^T: 35 KB, 108 VGPRs: four workgroups per CU); the online softmax
// state lives in registers across chunks as across the key tiles of a chunk.  At S <= 128 that is one workgroup and one chunk
// per (sequence, head).  At S = 512 (the reference's default --max_seq_length, retrieval/config.py:25) the four workgroups of a
// (sequence, head) re-read its K / V (128 KB) from the L2 of ONE XCD: blockIdx -> (XCD, pair, query chunk) keeps them on
// the same XCD, adjacent in dispatch order.  (Rounds 1-4 kept ALL keys of the sequence in LDS: 157 KB at S = 512 -- one
// workgroup per CU, one wave per SIMD, nothing to overlap the staging, the softmax arithmetic and the MFMAs with: 194 us per
// layer at 64 x 512 x 12 heads against 106 us in this form; 108 against 91 us at 512 x 128.)
constexpr int kLongChunk = 128;
constexpr float kExpScale = 0.125f * 1.4426950408889634f;   // log2(e) / sqrt(head_dim)
__global__ __launch_bounds__(256, 4) void attention_fwd(const _Float16* __restrict__ qkv, const _Float16* __restrict__ qkv_bias,
                                                          const int* __restrict__ seq_lens, const int* __restrict__ cu_seqlens,
                                                          int seq_len, int n_heads, int n_pairs, int n_qc,
                                                          _Float16* __restrict__ ctx) {
  constexpr int vt_stride = kLongChunk + kVtPad;
  __shared__ __attribute__((aligned(16))) _Float16 smem[kLongChunk * kKStride + kHeadDim * vt_stride];
  _Float16* k_lds = smem;                             // [128][kKStride]
  _Float16* vt_lds = smem + kLongChunk * kKStride;    // [kHeadDim][128 + kVtPad]  (V transposed)
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const int qc = (int)(slot % (unsigned)n_qc);
  const int pair = (int)(slot / (unsigned)n_qc) * 8 + (int)xcd;
  if (pair >= n_pairs) return;
  const int b = pair / n_heads;
  const int head = pair - b * n_heads;
  const int hidden = n_heads * kHeadDim;
  const long long row_stride = 3ll * hidden;
  const long long tok0 = cu_seqlens ? cu_seqlens[b] : (long long)b * seq_len;
  const _Float16* base = qkv + tok0 * row_stride + head * kHeadDim;
  int len = cu_seqlens ? cu_seqlens[b + 1] - cu_seqlens[b] : (seq_lens ? seq_lens[b] : seq_len);
  len = len < 1 ? 1 : (len > seq_len ? seq_len : len);
  const int rows_avail = cu_seqlens ? len : seq_len;   // token rows of this sequence that exist in memory
  if (qc * kLongChunk >= rows_avail) return;           // (the whole workgroup: no barrier has been met)
  const int n_ktiles = (len + 31) >> 5;
  const int n_kchunks = (n_ktiles + 3) >> 2;
  const _Float16* bias_q = qkv_bias ? qkv_bias + head * kHeadDim : nullptr;
  const _Float16* bias_v = qkv_bias ? qkv_bias + 2 * hidden + head * kHeadDim : nullptr;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int half = lane >> 5;
  const int qb = qc * 4 + wave;
  const bool active = qb * 32 < rows_avail;            // wave-uniform: a wave without queries still stages and meets barriers
  const int q = qb * 32 + li;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f16x8 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = active && q < rows_avail ? *(const f16x8*)(base + q * row_stride + (2 * j + half) * 8) : zero8;
    if (bias_q) qf[j] = qf[j] + *(const f16x8*)(bias_q + (2 * j + half) * 8);   // (key bias dropped, value bias on the output: see above)
  }
  float m = -__builtin_inff();
  float l = 0.f;
  f32x16 o0 = {0}, o1 = {0};

  // the K / V pieces of a chunk travel through registers: all eight loads of a thread are requested back to back and waited
  // for once (written in one loop with the LDS stores, hipcc waited per piece: 102 against 91 us at 512 x 128).  Requesting
  // chunk kc + 1 before the arithmetic of chunk kc (32 more live registers: three workgroups per CU) measured 2 % SLOWER
  // than four workgroups per CU hiding each other's round trips (ABLATIONS R5.11)
  constexpr int kIters = kLongChunk * 8 / 256;
  f16x8 kreg[kIters], vreg[kIters];
  auto request = [&](int kc) {
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int i = tid + it * 256;
      const int row = kc * kLongChunk + (i >> 3), c = i & 7;
      kreg[it] = zero8;
      vreg[it] = zero8;
      if (row < rows_avail) {
        const _Float16* src = base + row * row_stride + c * 8;
        kreg[it] = *(const f16x8*)(src + hidden);
        vreg[it] = *(const f16x8*)(src + 2 * hidden);
      }
    }
  };
  for (int kc = 0; kc < n_kchunks; ++kc) {
    request(kc);
    if (kc) __syncthreads();                           // every wave is done with the previous chunk
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int i = tid + it * 256;
      const int r = i >> 3, c = i & 7;
      const f16x8 kv = kreg[it], vv = vreg[it];
      *(f16x8*)(k_lds + r * kKStride + c * 8) = kv;
      // V^T[d][key]: rows r and r ^ 1 sit in lanes 8 apart (DPP row_ror:8).  The even row's lane writes d = 8c .. 8c+3,
      // the odd row's d = 8c+4 .. 8c+7, each as four 4-byte stores {key r & ~1, key r | 1} -- half the store instructions
      // of a 2-byte scatter and a quarter of its bank conflicts
      {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 w = __builtin_bit_cast(u32x4, vv);
        const bool odd = (r & 1) != 0;
        const unsigned s0 = odd ? w[0] : w[2], s1 = odd ? w[1] : w[3];
        const unsigned g0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xf, 0xf, false);
        const unsigned g1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xf, 0xf, false);
        const unsigned lo[2] = {odd ? g0 : w[0], odd ? g1 : w[1]};   // the d values of the EVEN key (low half of the store)
        const unsigned hi[2] = {odd ? w[2] : g0, odd ? w[3] : g1};   // ... of the odd key
        _Float16* dst = vt_lds + (c * 8 + (odd ? 4 : 0)) * vt_stride + (r & ~1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          *(unsigned*)(dst + (2 * t) * vt_stride) = __builtin_amdgcn_perm(hi[t], lo[t], 0x05040100u);
          *(unsigned*)(dst + (2 * t + 1) * vt_stride) = __builtin_amdgcn_perm(hi[t], lo[t], 0x07060302u);
        }
      }
    }
    __syncthreads();
    if (!active) continue;
    const int tiles_here = n_ktiles - kc * 4 < 4 ? n_ktiles - kc * 4 : 4;
    for (int kt = 0; kt < tiles_here; ++kt) {
      f32x16 st = {0};
      const _Float16* krow = k_lds + (kt * 32 + li) * kKStride + half * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x8 kf = *(const f16x8*)(krow + j * 16);
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[j], st, 0, 0, 0);
      }
      // the scores stay unscaled: exp((s - m) / 8) = exp2(s c - m c), c = log2(e) / 8 -- one fused multiply-add and one
      // v_exp_f32 per score (the scaling, the subtraction and __expf's own multiply by log2 e were three)
      const int key_base = (kc * 4 + kt) * 32;
      if (key_base + 32 > len) {                       // wave-uniform: only the last tile of a sequence is masked
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_base + (r & 3) + 8 * (r >> 2) + 4 * half;
          st[r] = key < len ? st[r] : -__builtin_inff();
        }
      }
      float mt = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = __builtin_fmaxf(mt, st[r]);
      mt = __builtin_fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = __builtin_fmaxf(m, mt);      // finite: key 0 is always valid
      const float mc = m_new * kExpScale;
      float rs = 0.f;
      f16x8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], kExpScale, -mc));
        rs += p;
        pf[r >> 3][r & 7] = (_Float16)p;
      }
      rs += __shfl_xor(rs, 32, 64);
      if (__any(m_new != m)) {                         // the running maximum moved for some query of the wave: rescale
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * kExpScale);
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o0[r] *= alpha;
          o1[r] *= alpha;
        }
      }
      l += rs;
      m = m_new;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int key0 = kt * 32 + 16 * jj + 4 * half;
        const _Float16* r0 = vt_lds + li * vt_stride + key0;
        const _Float16* r1 = vt_lds + (32 + li) * vt_stride + key0;
        f16x8 v0, v1;
        const f16x4 a0 = *(const f16x4*)r0, a1 = *(const f16x4*)(r0 + 8);
        const f16x4 b0 = *(const f16x4*)r1, b1 = *(const f16x4*)(r1 + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = a0[e];
          v0[4 + e] = a1[e];
          v1[e] = b0[e];
          v1[4 + e] = b1[e];
        }
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf[jj], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf[jj], o1, 0, 0, 0);
      }
    }
  }
  __syncthreads();                                     // the last chunk is dead: its K rows become the output staging tiles
  if (!active) return;
  _Float16* out_lds = smem + wave * 32 * kOutStride;   // 4 x 32 x 72 fp16 = the K chunk's 18 KB
  const float inv = 1.0f / l;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f16x4 a, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (_Float16)(o0[g * 4 + e] * inv);
      c[e] = (_Float16)(o1[g * 4 + e] * inv);
    }
    *(f16x4*)(out_lds + li * kOutStride + g * 8 + 4 * half) = a;
    *(f16x4*)(out_lds + li * kOutStride + 32 + g * 8 + 4 * half) = c;
  }
  // same wave writes and reads its tile: LDS ops of a wave complete in order
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), piece = lane & 7;   // 8 lanes x 16 B = one 128-byte row
    const int qq = qb * 32 + row;
    f16x8 v = *(const f16x8*)(out_lds + row * kOutStride + piece * 8);
    if (bias_v) v = v + *(const f16x8*)(bias_v + piece * 8);
    if (qq < rows_avail) *(f16x8*)(ctx + (tok0 + qq) * hidden + head * kHeadDim + piece * 8) = v;
  }
}

// Attention of the [CLS] query only (row 0 of every sequence): the last encoder layer feeds nothing
// but h[:, 0] to the pooler (retriever.py:41-42 takes BertModel's pooled output), so its attention
// output is needed for one query per sequence.  One wave per (sequence, head): lanes own keys for
// q.K^T (fp32), the softmax is a wave reduction, then lanes own the 64 output dims for P.V with the
// probabilities broadcast from LDS.  Streams K and V once: B*S*2*128 B per head, HBM-bound.
__global__ __launch_bounds__(256) void attention_cls_fwd(const _Float16* __restrict__ qkv,
                                                         const _Float16* __restrict__ qkv_bias,
                                                         const int* __restrict__ seq_lens,
                                                         const int* __restrict__ cu_seqlens, int seq_len,
                                                         int n_heads, int n_pairs, _Float16* __restrict__ ctx_cls) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  float* p_lds = (float*)smem + (threadIdx.x >> 6) * seq_len;   // [4 waves][seq_len] probabilities
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int b = pair / n_heads;
  const int head = pair - b * n_heads;
  const int hidden = n_heads * kHeadDim;
  const long long row_stride = 3ll * hidden;
  const long long tok0 = cu_seqlens ? cu_seqlens[b] : (long long)b * seq_len;
  const _Float16* base = qkv + tok0 * row_stride + head * kHeadDim;
  int len = cu_seqlens ? cu_seqlens[b + 1] - cu_seqlens[b] : (seq_lens ? seq_lens[b] : seq_len);
  len = len < 1 ? 1 : (len > seq_len ? seq_len : len);

  // 8 lanes per key: lane (g = lane>>3, c = lane&7) owns the 16-byte piece c of keys g, g+8, ... so that a
  // wave-instruction reads 8 whole 128-byte rows
  const int g = lane >> 3, c = lane & 7;
  float qv[8];
  {
    const f16x8 q8 = add_bias8(*(const f16x8*)(base + c * 8), qkv_bias ? qkv_bias + head * kHeadDim + c * 8 : nullptr);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = (float)q8[e];
  }
  float m = -__builtin_inff();
  for (int key0 = 0; key0 < len; key0 += 8) {
    const int key = key0 + g;
    float acc = 0.f;
    if (key < len) {
      const f16x8 k8 = add_bias8(*(const f16x8*)(base + key * row_stride + hidden + c * 8),
                                 qkv_bias ? qkv_bias + hidden + head * kHeadDim + c * 8 : nullptr);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += qv[e] * (float)k8[e];
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    acc *= 0.125f;
    if (key < len) {
      if (c == 0) p_lds[key] = acc;
      m = __builtin_fmaxf(m, acc);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
  float l = 0.f;
  // same wave wrote p_lds: LDS ops of a wave complete in order
  for (int key = lane; key < len; key += 64) {
    const float p = __expf(p_lds[key] - m);
    p_lds[key] = p;
    l += p;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
  // P.V with the same lane->(key group, piece) map: 8 partial outputs per dim, reduced over the groups
  float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int key0 = 0; key0 < len; key0 += 8) {
    const int key = key0 + g;
    if (key < len) {
      const float p = p_lds[key];
      const f16x8 v8 = add_bias8(*(const f16x8*)(base + key * row_stride + 2 * hidden + c * 8),
                                 qkv_bias ? qkv_bias + 2 * hidden + head * kHeadDim + c * 8 : nullptr);
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] += p * (float)v8[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o8[e] += __shfl_xor(o8[e], 8, 64);
    o8[e] += __shfl_xor(o8[e], 16, 64);
    o8[e] += __shfl_xor(o8[e], 32, 64);
  }
  if (g == 0) {
    const float inv = 1.0f / l;
    f16x8 out8;
#pragma unroll
    for (int e = 0; e < 8; ++e) out8[e] = (_Float16)(o8[e] * inv);
    *(f16x8*)(ctx_cls + (long long)b * hidden + head * kHeadDim + c * 8) = out8;
  }
}

}  // namespace
}  // namespace proqa

namespace proqa {

int launch_attention(const void* qkv, const void* qkv_bias, const int32_t* seq_lens_dev, const int32_t* cu_seqlens_dev,
                     int batch, int seq_len, int n_heads, void* ctx_out, void* stream) {
  if (!qkv || !ctx_out) return fail(PROQA_EINVAL, "attention: NULL argument");
  if (batch < 0 || seq_len <= 0 || n_heads <= 0) return fail(PROQA_EINVAL, "attention: bad sizes");
  if (batch == 0) return PROQA_OK;
  const int n_pairs = batch * n_heads, n_qc = (seq_len + kLongChunk - 1) / kLongChunk;
  const unsigned grid = (unsigned)((n_pairs + 7) / 8 * 8) * (unsigned)n_qc;
  hipLaunchKernelGGL(attention_fwd, dim3(grid), dim3(256), 0, as_stream(stream), (const _Float16*)qkv, (const _Float16*)qkv_bias,
                     (const int*)seq_lens_dev, (const int*)cu_seqlens_dev, seq_len, n_heads, n_pairs, n_qc, (_Float16*)ctx_out);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

int launch_attention_cls(const void* qkv, const void* qkv_bias, const int32_t* seq_lens_dev,
                         const int32_t* cu_seqlens_dev, int batch, int seq_len, int n_heads, void* ctx_cls_out,
                         void* stream) {
  if (!qkv || !ctx_cls_out) return fail(PROQA_EINVAL, "attention_cls: NULL argument");
  if (batch < 0 || seq_len <= 0 || n_heads <= 0) return fail(PROQA_EINVAL, "attention_cls: bad sizes");
  const size_t lds = (size_t)4 * seq_len * sizeof(float);
  if (lds > 64 * 1024) return fail(PROQA_EINVAL, "attention_cls: seq_len=%d too long", seq_len);
  if (batch == 0) return PROQA_OK;
  const int n_pairs = batch * n_heads;
  hipLaunchKernelGGL(attention_cls_fwd, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), lds, as_stream(stream),
                     (const _Float16*)qkv, (const _Float16*)qkv_bias, (const int*)seq_lens_dev,
                     (const int*)cu_seqlens_dev, seq_len, n_heads, n_pairs, (_Float16*)ctx_cls_out);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

}  // namespace proqa

using namespace proqa;

extern "C" int proqa_attention_f16(const void* qkv, const int32_t* seq_lens_dev, int batch, int seq_len,
                                   int n_heads, void* ctx_out, void* stream) {
  return launch_attention(qkv, nullptr, seq_lens_dev, nullptr, batch, seq_len, n_heads, ctx_out, stream);
}

extern "C" int proqa_attention_varlen_f16(const void* qkv_packed, const int32_t* cu_seqlens_dev, int batch,
                                          int max_seq_len, int n_heads, void* ctx_packed_out, void* stream) {
  if (!cu_seqlens_dev) return fail(PROQA_EINVAL, "attention_varlen: NULL cu_seqlens");
  return launch_attention(qkv_packed, nullptr, nullptr, cu_seqlens_dev, batch, max_seq_len, n_heads, ctx_packed_out, stream);
}

extern "C" int proqa_attention_cls_f16(const void* qkv, const int32_t* seq_lens_dev, int batch, int seq_len,
                                       int n_heads, void* ctx_cls_out, void* stream) {
  return launch_attention_cls(qkv, nullptr, seq_lens_dev, nullptr, batch, seq_len, n_heads, ctx_cls_out, stream);
}

extern "C" int proqa_attention_cls_varlen_f16(const void* qkv_packed, const int32_t* cu_seqlens_dev, int batch,
                                              int max_seq_len, int n_heads, void* ctx_cls_out, void* stream) {
  if (!cu_seqlens_dev) return fail(PROQA_EINVAL, "attention_cls_varlen: NULL cu_seqlens");
  return launch_attention_cls(qkv_packed, nullptr, nullptr, cu_seqlens_dev, batch, max_seq_len, n_heads, ctx_cls_out, stream);
}

extern "C" int proqa_attention_ex_f16(const void* qkv, const void* qkv_bias, const int32_t* seq_lens_dev,
                                      const int32_t* cu_seqlens_dev, int batch, int seq_len, int n_heads, int cls_only,
                                      void* out, void* stream) {
  if ((seq_lens_dev != nullptr) == (cu_seqlens_dev != nullptr))
    return fail(PROQA_EINVAL, "attention_ex: pass exactly one of seq_lens (padded layout) and cu_seqlens (packed layout)");
  return cls_only ? launch_attention_cls(qkv, qkv_bias, seq_lens_dev, cu_seqlens_dev, batch, seq_len, n_heads, out, stream)
                  : launch_attention(qkv, qkv_bias, seq_lens_dev, cu_seqlens_dev, batch, seq_len, n_heads, out, stream);
}

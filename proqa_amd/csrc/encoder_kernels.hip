// Non-GEMM kernels of the BERT tower behind BertForRetriever.get_embed
// (/root/reference/retrieval/retriever.py:33-43; the arithmetic is transformers' BertModel:
// embeddings -> 12 x {self-attention, output dense+LN, FFN} -> pooler, then proj_{q,c}).
// The dense 768x768 / 768x3072 projections are rocBLAS GEMMs issued by encoder.cpp; everything
// between them is here.  Activations are fp16 in HBM, statistics fp32.
//
// All row-wise kernels use one wave64 per row with 16-byte (8 x fp16) accesses per lane: at
// hidden=768 a row is 96 such chunks, so lanes 0-31 hold two chunks and lanes 32-63 one; the
// row never leaves registers between the load and the normalised store.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace proqa {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxChunksPerLane = 2;  // hidden <= 64 lanes * 2 chunks * 8 = 1024
constexpr int kRowsPerBlock = 4;      // 4 waves per 256-thread block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LayerNorm over the register-resident row (two-pass in registers: mean, then variance of the
// deviations — matches torch.nn.LayerNorm's biased variance) and 16-byte stores.
__device__ __forceinline__ void layernorm_store(float (&x)[kMaxChunksPerLane][8], int lane, int n_chunks,
                                                int hidden, const _Float16* __restrict__ gamma,
                                                const _Float16* __restrict__ beta, float eps,
                                                _Float16* __restrict__ out_row) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunksPerLane; ++c)
    if (lane + 64 * c < n_chunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s += x[c][i];
    }
  const float mean = wave_sum(s) / (float)hidden;
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunksPerLane; ++c)
    if (lane + 64 * c < n_chunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = x[c][i] - mean;
        v += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(v) / (float)hidden + eps);
#pragma unroll
  for (int c = 0; c < kMaxChunksPerLane; ++c) {
    const int chunk = lane + 64 * c;
    if (chunk < n_chunks) {
      const f16x8 g = *(const f16x8*)(gamma + chunk * 8);
      const f16x8 b = *(const f16x8*)(beta + chunk * 8);
      f16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (_Float16)((x[c][i] - mean) * rstd * (float)g[i] + (float)b[i]);
      *(f16x8*)(out_row + chunk * 8) = o;
    }
  }
}

// BertEmbeddings: LN(word[id] + pos[s] + type[0])
// cu_seqlens != nullptr: ids stay padded [B, seq_len] but the output is packed -- token (b, s) with
// s < len(b) goes to row cu_seqlens[b] + s, padding positions produce nothing
__global__ __launch_bounds__(256) void embed_layernorm(const long long* __restrict__ ids, long long n_tokens,
                                                       const int* __restrict__ cu_seqlens, int seq_len, int hidden,
                                                       const _Float16* __restrict__ word, long long vocab,
                                                       const _Float16* __restrict__ pos,
                                                       const _Float16* __restrict__ type0,
                                                       const _Float16* __restrict__ gamma,
                                                       const _Float16* __restrict__ beta, float eps,
                                                       _Float16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long tok = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (tok >= n_tokens) return;
  long long id = ids[tok];
  if (id < 0 || id >= vocab) id = 0;  // never index outside the table
  const int s = (int)(tok % seq_len);
  long long out_row = tok;
  if (cu_seqlens) {
    const long long b = tok / seq_len;
    const int first = cu_seqlens[b];
    if (s >= cu_seqlens[b + 1] - first) return;
    out_row = first + s;
  }
  const int n_chunks = hidden >> 3;
  const _Float16* w = word + id * hidden;
  const _Float16* p = pos + (long long)s * hidden;
  float x[kMaxChunksPerLane][8];
#pragma unroll
  for (int c = 0; c < kMaxChunksPerLane; ++c) {
    const int chunk = lane + 64 * c;
    if (chunk < n_chunks) {
      const f16x8 a = *(const f16x8*)(w + chunk * 8);
      const f16x8 b = *(const f16x8*)(p + chunk * 8);
      const f16x8 t = *(const f16x8*)(type0 + chunk * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[c][i] = (float)a[i] + (float)b[i] + (float)t[i];
    }
  }
  layernorm_store(x, lane, n_chunks, hidden, gamma, beta, eps, out + out_row * hidden);
}

// BertSelfOutput / BertOutput: LN(dense_out + bias + residual)
__global__ __launch_bounds__(256) void bias_residual_layernorm(const _Float16* __restrict__ xin,
                                                               const _Float16* __restrict__ bias,
                                                               const _Float16* __restrict__ residual,
                                                               const _Float16* __restrict__ gamma,
                                                               const _Float16* __restrict__ beta, float eps,
                                                               long long rows, int cols,
                                                               _Float16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n_chunks = cols >> 3;
  const _Float16* xr = xin + row * cols;
  const _Float16* rr = residual + row * cols;
  float x[kMaxChunksPerLane][8];
#pragma unroll
  for (int c = 0; c < kMaxChunksPerLane; ++c) {
    const int chunk = lane + 64 * c;
    if (chunk < n_chunks) {
      const f16x8 a = *(const f16x8*)(xr + chunk * 8);
      const f16x8 r = *(const f16x8*)(rr + chunk * 8);
      const f16x8 b = *(const f16x8*)(bias + chunk * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[c][i] = (float)a[i] + (float)b[i] + (float)r[i];
    }
  }
  layernorm_store(x, lane, n_chunks, cols, gamma, beta, eps, out + row * cols);
}

// BertIntermediate: x = gelu(x + bias), exact erf form (hidden_act = 'gelu')
__global__ __launch_bounds__(256) void bias_gelu(_Float16* __restrict__ x, const _Float16* __restrict__ bias,
                                                 long long n_chunks_total, int chunks_per_row) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks_total; c += stride) {
    const int col_chunk = (int)(c % chunks_per_row);
    f16x8 v = *(f16x8*)(x + c * 8);
    const f16x8 b = *(const f16x8*)(bias + col_chunk * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = (float)v[i] + (float)b[i];
      v[i] = (_Float16)(0.5f * t * (1.0f + erff(t * 0.70710678118654752440f)));
    }
    *(f16x8*)(x + c * 8) = v;
  }
}

// BertPooler + projection head: out[b] = Wproj . tanh(Wp . h[b,0] + bp) + bproj, as two launches
// of one small MFMA kernel  Y[n][m] = act(sum_k W[m][k] X[n][k] + bias[m]).
// Orientation: features = M (A operand = rows of the [out,in] weight, read as contiguous 16-byte
// pieces), sequences = N, so every lane owns one sequence column and 16 features of it; a wave
// computes a 32x32 tile over the whole K, a workgroup 128 features x 32 sequences.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool TANH, typename OutT>
__global__ __launch_bounds__(256) void cls_dense_mfma(const _Float16* __restrict__ x, long long x_stride, int n_seq,
                                                      const _Float16* __restrict__ w, const _Float16* __restrict__ bias,
                                                      int n_feat, int k_dim, OutT* __restrict__ y, int y_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int n0 = blockIdx.y * 32;
  if (m0 >= n_feat) return;
  const int n = n0 + li;
  const _Float16* wrow = w + (long long)(m0 + li) * k_dim + half * 8;
  const _Float16* xrow = x + (long long)(n < n_seq ? n : n_seq - 1) * x_stride + half * 8;
  f32x16 acc = {0};
  for (int k = 0; k < k_dim; k += 16) {
    const f16x8 a = *(const f16x8*)(wrow + k);
    const f16x8 b = *(const f16x8*)(xrow + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  if (n >= n_seq) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int m = m0 + 8 * g + 4 * half;  // registers 4g..4g+3 hold features m..m+3
    const f16x4 bv = *(const f16x4*)(bias + m);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[4 * g + e] + (float)bv[e];
      if (TANH) v = tanhf(v);
      y[(long long)n * y_stride + m + e] = (OutT)v;
    }
  }
}

// Dense layer for a FEW token rows (a question is <= 30 tokens; up to 256 rows): y[n][m] = act(sum_k w[m][k] x[n][k] + bias[m]).
// The library GEMM costs ~20 us of host and device time per call on such shapes, whatever the row count (its macro-tiles
// are mostly padding), and a single-question encode is 48 of them.  Here one workgroup owns a 32-feature x 32-row output
// tile; its four waves split K in quarters (two accumulator chains each, so consecutive MFMAs do not wait for each other),
// reduce through LDS and write 8-byte pieces; N/32 workgroups stream the weight once.  ACT: 0 none, 1 erf GELU.
// bias may be null.  (A variant that issued all of a wave's loads before its first MFMA, with up to 16 waves over K, was
// not faster: at 16 rows the launch itself is the cost, ~5.6 us per kernel on a dependent chain.)
__device__ __forceinline__ float gelu_erf_small(float t) { return 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f)); }

template <int ACT>
__global__ __launch_bounds__(256) void small_dense_mfma(const _Float16* __restrict__ x, int n_rows,
                                                        const _Float16* __restrict__ w, const _Float16* __restrict__ bias,
                                                        int n_feat, int k_dim, _Float16* __restrict__ y) {
  __shared__ float part[4][32][33];   // [wave][feature][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int n = n0 + li;
  const int kq = k_dim / 4;           // this wave's K range (k_dim is a multiple of 128)
  const _Float16* wrow = w + (long long)(m0 + li) * k_dim + wave * kq + half * 8;
  const _Float16* xrow = x + (long long)(n < n_rows ? n : n_rows - 1) * k_dim + wave * kq + half * 8;
  f32x16 acc0 = {0}, acc1 = {0};
  for (int k = 0; k < kq; k += 32) {   // (hipcc declines to unroll this loop: no pragma)
    const f16x8 a0 = *(const f16x8*)(wrow + k), b0 = *(const f16x8*)(xrow + k);
    const f16x8 a1 = *(const f16x8*)(wrow + k + 16), b1 = *(const f16x8*)(xrow + k + 16);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc1, 0, 0, 0);
  }
  // lane (li, half) holds row n, features (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * half][li] = acc0[r] + acc1[r];
  __syncthreads();
  // 256 threads: row = tid / 8, features 4*(tid % 8) .. +3
  const int row = tid >> 3, f0 = (tid & 7) * 4;
  if (n0 + row >= n_rows) return;
  f16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = part[0][f0 + e][row] + part[1][f0 + e][row] + part[2][f0 + e][row] + part[3][f0 + e][row];
    if (bias) v += (float)bias[m0 + f0 + e];
    if (ACT == 1) v = gelu_erf_small(v);
    o[e] = (_Float16)v;
  }
  *(f16x4*)(y + (long long)(n0 + row) * n_feat + m0 + f0) = o;
}

}  // namespace
}  // namespace proqa

using namespace proqa;

// ---- helpers of the C++ encoder driver (encoder.cpp) ---------------------------------------
namespace proqa {
namespace {

// cu[0] = 0, cu[b+1] = cu[b] + clamp(lens[b], 1, ...): exclusive prefix of the sequence lengths (one block)
__global__ __launch_bounds__(256) void cu_seqlens_kernel(const int* __restrict__ lens, int batch, int* __restrict__ cu) {
  __shared__ int part[256];
  const int tid = threadIdx.x;
  const int per = (batch + 255) / 256;
  const int b0 = tid * per, b1 = b0 + per < batch ? b0 + per : batch;
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += lens[b] < 1 ? 1 : lens[b];
  part[tid] = sum;
  __syncthreads();
  int base = 0;
  for (int t = 0; t < tid; ++t) base += part[t];
  if (tid == 0) cu[0] = 0;
  for (int b = b0; b < b1; ++b) {
    base += lens[b] < 1 ? 1 : lens[b];
    cu[b + 1] = base;
  }
}

// dst[r] = row (index ? index[r] : r * fixed_stride_rows) of src; cols a multiple of 8 (16-byte pieces)
__global__ void gather_rows_kernel(const _Float16* __restrict__ src, long long row_stride, const int* __restrict__ index,
                                   long long fixed_stride_rows, int n_rows, int chunks_per_row, _Float16* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_rows * chunks_per_row) return;
  const int r = (int)(t / chunks_per_row), c = (int)(t - (long long)r * chunks_per_row);
  const long long row = index ? (long long)index[r] : (long long)r * fixed_stride_rows;
  *(f16x8*)(dst + (long long)r * chunks_per_row * 8 + c * 8) = *(const f16x8*)(src + row * row_stride + c * 8);
}

}  // namespace

int launch_cu_seqlens(const int32_t* seq_lens_dev, int batch, int32_t* cu_out, void* stream) {
  hipLaunchKernelGGL(cu_seqlens_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const int*)seq_lens_dev, batch,
                     (int*)cu_out);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

// y[rows, n_feat] = act(x[rows, k] . w[n_feat, k]^T + bias) for a few rows; n_feat % 32 == 0, k % 128 == 0
int launch_small_dense(const void* x, int rows, const void* w, const void* bias, int n_feat, int k_dim, int act, void* y,
                       void* stream) {
  if (rows == 0) return PROQA_OK;
  if (n_feat % 32 || k_dim % 128 || k_dim <= 0) return fail(PROQA_EINVAL, "small_dense: n_feat=%d k=%d", n_feat, k_dim);
  const dim3 g((unsigned)(n_feat / 32), (unsigned)((rows + 31) / 32));
  if (act == 1)
    hipLaunchKernelGGL(small_dense_mfma<1>, g, dim3(256), 0, as_stream(stream), (const _Float16*)x, rows, (const _Float16*)w,
                       (const _Float16*)bias, n_feat, k_dim, (_Float16*)y);
  else
    hipLaunchKernelGGL(small_dense_mfma<0>, g, dim3(256), 0, as_stream(stream), (const _Float16*)x, rows, (const _Float16*)w,
                       (const _Float16*)bias, n_feat, k_dim, (_Float16*)y);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

int launch_gather_rows(const void* src, int64_t src_row_stride_elems, const int32_t* row_index_dev, int64_t fixed_stride_rows,
                       int n_rows, int cols, void* dst, void* stream) {
  if (n_rows == 0) return PROQA_OK;
  const long long n = (long long)n_rows * (cols / 8);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const _Float16*)src, (long long)src_row_stride_elems, (const int*)row_index_dev,
                     (long long)fixed_stride_rows, n_rows, cols / 8, (_Float16*)dst);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

}  // namespace proqa

extern "C" {

static int launch_embed_layernorm(const int64_t* ids_dev, int64_t n_tokens, const int32_t* cu_seqlens_dev, int seq_len,
                                  int hidden, const void* word_emb, int64_t vocab, const void* pos_emb,
                                  const void* type_emb, const void* ln_gamma, const void* ln_beta, float eps, void* out,
                                  void* stream) {
  if (!ids_dev || !word_emb || !pos_emb || !type_emb || !ln_gamma || !ln_beta || !out)
    return fail(PROQA_EINVAL, "embed_layernorm: NULL argument");
  if (n_tokens < 0 || seq_len <= 0 || vocab <= 0) return fail(PROQA_EINVAL, "embed_layernorm: bad sizes");
  if (hidden <= 0 || hidden % 8 || hidden > 64 * kMaxChunksPerLane * 8)
    return fail(PROQA_EINVAL, "embed_layernorm: hidden=%d must be a multiple of 8 and <= %d", hidden,
                64 * kMaxChunksPerLane * 8);
  if (n_tokens == 0) return PROQA_OK;
  const unsigned grid = (unsigned)ceil_div<int64_t>(n_tokens, kRowsPerBlock);
  hipLaunchKernelGGL(embed_layernorm, dim3(grid), dim3(256), 0, as_stream(stream), (const long long*)ids_dev,
                     (long long)n_tokens, (const int*)cu_seqlens_dev, seq_len, hidden, (const _Float16*)word_emb,
                     (long long)vocab, (const _Float16*)pos_emb, (const _Float16*)type_emb, (const _Float16*)ln_gamma,
                     (const _Float16*)ln_beta, eps, (_Float16*)out);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

int proqa_embed_layernorm_f16(const int64_t* ids_dev, int64_t n_tokens, int seq_len, int hidden,
                              const void* word_emb, int64_t vocab, const void* pos_emb, const void* type_emb,
                              const void* ln_gamma, const void* ln_beta, float eps, void* out, void* stream) {
  return launch_embed_layernorm(ids_dev, n_tokens, nullptr, seq_len, hidden, word_emb, vocab, pos_emb, type_emb,
                                ln_gamma, ln_beta, eps, out, stream);
}

int proqa_embed_layernorm_varlen_f16(const int64_t* ids_dev, const int32_t* cu_seqlens_dev, int batch, int seq_len,
                                     int hidden, const void* word_emb, int64_t vocab, const void* pos_emb,
                                     const void* type_emb, const void* ln_gamma, const void* ln_beta, float eps,
                                     void* out_packed, void* stream) {
  if (!cu_seqlens_dev || batch < 0) return fail(PROQA_EINVAL, "embed_layernorm_varlen: bad argument");
  return launch_embed_layernorm(ids_dev, (int64_t)batch * seq_len, cu_seqlens_dev, seq_len, hidden, word_emb, vocab,
                                pos_emb, type_emb, ln_gamma, ln_beta, eps, out_packed, stream);
}

int proqa_bias_residual_layernorm_f16(const void* x, const void* bias, const void* residual, const void* gamma,
                                      const void* beta, float eps, int64_t rows, int cols, void* out,
                                      void* stream) {
  if (!x || !bias || !residual || !gamma || !beta || !out)
    return fail(PROQA_EINVAL, "bias_residual_layernorm: NULL argument");
  if (rows < 0 || cols <= 0 || cols % 8 || cols > 64 * kMaxChunksPerLane * 8)
    return fail(PROQA_EINVAL, "bias_residual_layernorm: cols=%d must be a multiple of 8 and <= %d", cols,
                64 * kMaxChunksPerLane * 8);
  if (rows == 0) return PROQA_OK;
  const unsigned grid = (unsigned)ceil_div<int64_t>(rows, kRowsPerBlock);
  hipLaunchKernelGGL(bias_residual_layernorm, dim3(grid), dim3(256), 0, as_stream(stream), (const _Float16*)x,
                     (const _Float16*)bias, (const _Float16*)residual, (const _Float16*)gamma,
                     (const _Float16*)beta, eps, (long long)rows, cols, (_Float16*)out);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

int proqa_bias_gelu_f16(void* x, const void* bias, int64_t rows, int cols, void* stream) {
  if (!x || !bias) return fail(PROQA_EINVAL, "bias_gelu: NULL argument");
  if (rows < 0 || cols <= 0 || cols % 8) return fail(PROQA_EINVAL, "bias_gelu: cols=%d must be a multiple of 8", cols);
  if (rows == 0) return PROQA_OK;
  const long long n_chunks = rows * (long long)(cols / 8);
  const long long want = ceil_div<long long>(n_chunks, 256);
  const unsigned grid = (unsigned)std::min<long long>(want, (long long)device_cu_count() * 8);
  hipLaunchKernelGGL(bias_gelu, dim3(grid), dim3(256), 0, as_stream(stream), (_Float16*)x, (const _Float16*)bias,
                     n_chunks, cols / 8);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

int proqa_pool_project_f16(const void* h, int batch, int seq_len, int hidden, const void* w_pool,
                           const void* b_pool, const void* w_proj, const void* b_proj, void* pooled_ws, void* out,
                           int out_dtype, void* stream) {
  if (!h || !w_pool || !b_pool || !w_proj || !b_proj || !pooled_ws || !out)
    return fail(PROQA_EINVAL, "pool_project: NULL argument");
  if (batch < 0 || seq_len <= 0 || hidden <= 0 || hidden % 32)
    return fail(PROQA_EINVAL, "pool_project: hidden=%d must be a positive multiple of 32", hidden);
  if (out_dtype != PROQA_F16 && out_dtype != PROQA_F32) return fail(PROQA_EINVAL, "pool_project: bad out dtype");
  if (batch == 0) return PROQA_OK;
  hipStream_t st = as_stream(stream);
  const dim3 blk(256);
  const unsigned ny = (unsigned)ceil_div<int>(batch, 32);
  // pooled[b] = tanh(Wp . h[b,0] + bp): the CLS rows are rows of h at stride seq_len*hidden
  hipLaunchKernelGGL((cls_dense_mfma<true, _Float16>), dim3((unsigned)ceil_div<int>(hidden, 128), ny), blk, 0, st,
                     (const _Float16*)h, (long long)seq_len * hidden, batch, (const _Float16*)w_pool,
                     (const _Float16*)b_pool, hidden, hidden, (_Float16*)pooled_ws, hidden);
  PROQA_LAUNCH_CHECK();
  const dim3 g2((unsigned)ceil_div<int>(PROQA_EMBED_DIM, 128), ny);
  if (out_dtype == PROQA_F16)
    hipLaunchKernelGGL((cls_dense_mfma<false, _Float16>), g2, blk, 0, st, (const _Float16*)pooled_ws, (long long)hidden,
                       batch, (const _Float16*)w_proj, (const _Float16*)b_proj, PROQA_EMBED_DIM, hidden,
                       (_Float16*)out, PROQA_EMBED_DIM);
  else
    hipLaunchKernelGGL((cls_dense_mfma<false, float>), g2, blk, 0, st, (const _Float16*)pooled_ws, (long long)hidden,
                       batch, (const _Float16*)w_proj, (const _Float16*)b_proj, PROQA_EMBED_DIM, hidden, (float*)out,
                       PROQA_EMBED_DIM);
  PROQA_LAUNCH_CHECK();
  return PROQA_OK;
}

}  // extern "C"

// BertForRetriever.get_embed behind the C ABI (proqa_encoder_* in proqa_hip.h): the whole tower --
// embeddings, N encoder layers, pooler, projection -- driven from C++ on caller-owned fp16 weights.
// Replaces /root/reference/retrieval/retriever.py:33-43 (+ transformers BertModel).  The dense layers are
// rocBLAS GEMMs (rocblas_gemm_ex, fp16 in / fp32 accumulate; the library resolves them against the
// rocBLAS of the process, see proqa_amd/_lib.py); everything else is the HIP kernels of this library.
#define ROCBLAS_BETA_FEATURES_API   // rocblas_gemm_ex_get_solutions (opt-in tuning only)
#include <rocblas/rocblas.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "attention.h"
#include "common.h"
#include "lt_gemm.h"

using namespace proqa;

namespace proqa {
// encoder_kernels.hip
int launch_cu_seqlens(const int32_t* seq_lens_dev, int batch, int32_t* cu_out, void* stream);
int launch_small_dense(const void* x, int rows, const void* w, const void* bias, int n_feat, int k_dim, int act, void* y,
                       void* stream);
int launch_gather_rows(const void* src, int64_t src_row_stride_elems, const int32_t* row_index_dev, int64_t fixed_stride_rows,
                       int n_rows, int cols, void* dst, void* stream);
}  // namespace proqa

namespace {

constexpr size_t kMaxTunedShapes = 32;
// Token rows handed to the GEMMs are a multiple of this (when large).  The library's default kernel for these shapes is
// a 256 x 256 macro-tile scheme that runs 25-40 % slower on an ODD number of 256-row tiles (scripts/dev_gemm_vs_m.py:
// N=768 K=3072 at M = 161 x 256: 998 TFLOP/s, at 162 x 256: 1373), so packed (ragged) batches are padded to 512.
constexpr int kRowTile = 512;

struct Workspace {
  void* base = nullptr;
  size_t bytes = 0;
  int64_t rows = 0;   // token rows it was sized for
  int batch = 0;
  _Float16 *h = nullptr, *h1 = nullptr, *qkv = nullptr, *ctx = nullptr, *tmp = nullptr, *ff = nullptr;
  _Float16 *c_ctx = nullptr, *c_res = nullptr, *c_tmp = nullptr, *c_h1 = nullptr, *c_ff = nullptr, *c_h = nullptr,
           *pooled = nullptr;
  int32_t* cu = nullptr;
};

}  // namespace

struct proqa_encoder {
  proqa_bert_weights w;
  std::vector<proqa_bert_layer> layers;
  rocblas_handle blas = nullptr;
  LtGemm* lt = nullptr;   // large products on a hipBLASLt kernel pinned by name (lt_gemm.cpp); nullptr: rocblas_gemm_ex for everything
  Workspace ws;
  int device = 0;
  // opt-in GEMM solution tuning (proqa_encoder_set_gemm_tuning): shape -> rocBLAS solution index (0 = default)
  bool own_ffn1 = true;   // BertIntermediate on the hand-written GEMM with the fused bias + GELU epilogue (PROQA_FFN1=lib: library GEMM + bias_gelu)
  bool tune = false;
  std::map<std::tuple<int64_t, int, int>, int> solution;
};

namespace {

int blas_fail(rocblas_status s, const char* what) {
  return fail(PROQA_EHIP, "%s failed: rocblas status %d", what, (int)s);
}
#define PROQA_BLAS(call)                                   \
  do {                                                     \
    rocblas_status _s = (call);                            \
    if (_s != rocblas_status_success) return blas_fail(_s, #call); \
  } while (0)

// out[M,N] = x[M,K] . w[N,K]^T, row-major fp16, fp32 accumulate.  Column-major view: out'[N,M] = w'^T x'.
rocblas_status gemm_call(proqa_encoder* e, const _Float16* x, const void* w, _Float16* out, int64_t M, int N, int K,
                         int solution) {
  const float alpha = 1.0f, beta = 0.0f;
  return rocblas_gemm_ex(e->blas, rocblas_operation_transpose, rocblas_operation_none, N, (rocblas_int)M, K, &alpha, w,
                         rocblas_datatype_f16_r, K, x, rocblas_datatype_f16_r, K, &beta, out, rocblas_datatype_f16_r, N, out,
                         rocblas_datatype_f16_r, N, rocblas_datatype_f32_r,
                         solution ? rocblas_gemm_algo_solution_index : rocblas_gemm_algo_standard, solution, 0);
}

// Median-of-`reps` time (ms) of `calls` back-to-back launches of one solution on the encoder's stream.
float time_solution(proqa_encoder* e, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const _Float16* x, const void* w,
                    _Float16* out, int64_t M, int N, int K, int solution, int reps, int calls) {
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0, st);
    for (int c = 0; c < calls; ++c)
      if (gemm_call(e, x, w, out, M, N, K, solution) != rocblas_status_success) return 1e30f;
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms / calls);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wdeprecated-declarations"   // the solution-listing call is a rocBLAS "beta" API
// Opt-in: the first time a large shape is seen, every solution rocBLAS lists for it is timed on the real operands
// (the output buffer is overwritten by the real call afterwards) and the winner replaces the library's default
// only if an interleaved re-match confirms >= 2 % (measurement noise is of that order).
int tune_shape(proqa_encoder* e, hipStream_t st, const _Float16* x, const void* w, _Float16* out, int64_t M, int N, int K) {
  const float alpha = 1.0f, beta = 0.0f;
  rocblas_int n = 0;
  if (rocblas_gemm_ex_get_solutions(e->blas, rocblas_operation_transpose, rocblas_operation_none, N, (rocblas_int)M, K, &alpha,
                                    w, rocblas_datatype_f16_r, K, x, rocblas_datatype_f16_r, K, &beta, out,
                                    rocblas_datatype_f16_r, N, out, rocblas_datatype_f16_r, N, rocblas_datatype_f32_r,
                                    rocblas_gemm_algo_solution_index, 0, nullptr, &n) != rocblas_status_success || n <= 0)
    return 0;
  std::vector<rocblas_int> sols(n);
  if (rocblas_gemm_ex_get_solutions(e->blas, rocblas_operation_transpose, rocblas_operation_none, N, (rocblas_int)M, K, &alpha,
                                    w, rocblas_datatype_f16_r, K, x, rocblas_datatype_f16_r, K, &beta, out,
                                    rocblas_datatype_f16_r, N, out, rocblas_datatype_f16_r, N, rocblas_datatype_f32_r,
                                    rocblas_gemm_algo_solution_index, 0, sols.data(), &n) != rocblas_status_success)
    return 0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 0;
  int best = 0;
  float best_ms = 1e30f;
  for (rocblas_int s : sols) {
    const float ms = time_solution(e, st, e0, e1, x, w, out, M, N, K, s, 3, 4);
    if (ms < best_ms) {
      best_ms = ms;
      best = s;
    }
  }
  // re-match against the default, interleaved
  float d = 0.f, b = 0.f;
  for (int r = 0; r < 5 && best; ++r) {
    d += time_solution(e, st, e0, e1, x, w, out, M, N, K, 0, 1, 8);
    b += time_solution(e, st, e0, e1, x, w, out, M, N, K, best, 1, 8);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (best && b < 0.98f * d) ? best : 0;
}
#pragma clang diagnostic pop

constexpr int kSmallDenseMaxRows = 256;   // up to here the dense layers run on small_dense_mfma (encoder_kernels.hip)

const int kSmallDenseRows = [] { const char* v = getenv("PROQA_SMALL_DENSE"); return v ? atoi(v) : kSmallDenseMaxRows; }();   // developer A/B switch (0: off)

inline bool small_dense_ok(int64_t M, int N, int K) { return M <= kSmallDenseRows && N % 32 == 0 && K % 128 == 0; }

int gemm_tn(proqa_encoder* e, const _Float16* x, const void* w, _Float16* out, int64_t M, int N, int K, hipStream_t st) {
  if (M == 0) return PROQA_OK;
  if (small_dense_ok(M, N, K)) return launch_small_dense(x, (int)M, w, nullptr, N, K, 0, out, st);
  if (e->lt && M >= 4096) {   // (the pinned kernel is a 256 x 256 macro-tile scheme: large token counts only)
    const int r = lt_gemm_tn(e->lt, x, w, out, M, N, K, st);
    if (r <= 0) return r;     // launched, or failed; 1 = this library / shape has no pinned kernel: rocBLAS below
  }
  int solution = 0;
  if (e->tune && M >= 4096) {
    const auto key = std::make_tuple(M, N, K);
    auto it = e->solution.find(key);
    // at most kMaxTunedShapes shapes are tuned per encoder (~0.3-1 s each): corpora with ever-changing packed
    // row counts fall back to the library's default once the budget is spent
    if (it == e->solution.end() && e->solution.size() < kMaxTunedShapes)
      it = e->solution.emplace(key, tune_shape(e, st, x, w, out, M, N, K)).first;
    solution = it == e->solution.end() ? 0 : it->second;
  }
  PROQA_BLAS(gemm_call(e, x, w, out, M, N, K, solution));
  return PROQA_OK;
}

int ensure_workspace(proqa_encoder* e, int batch, int64_t rows, hipStream_t st) {
  Workspace& ws = e->ws;
  if (rows <= ws.rows && batch <= ws.batch) return PROQA_OK;
  rows = std::max(rows, ws.rows);
  batch = std::max(batch, ws.batch);
  const int H = e->w.hidden, I = e->w.intermediate;
  auto al = [](size_t b) { return round_up<size_t>(b, 256); };
  const size_t tok = (size_t)rows, bt = (size_t)batch;
  const size_t sizes[] = {al(tok * H * 2),     al(tok * H * 2), al(tok * 3 * H * 2), al(tok * H * 2), al(tok * H * 2),
                          al(tok * I * 2),     al(bt * H * 2),  al(bt * H * 2),      al(bt * H * 2),  al(bt * H * 2),
                          al(bt * I * 2),      al(bt * H * 2),  al(bt * H * 2),      al((bt + 1) * 4)};
  size_t total = 0;
  for (size_t s : sizes) total += s;
  if (ws.base) PROQA_HIP(hipFree(ws.base));
  ws = Workspace();
  hipError_t err = hipMalloc(&ws.base, total);
  if (err != hipSuccess) return fail(PROQA_ENOMEM, "encoder workspace of %zu B: %s", total, hipGetErrorString(err));
  // rows past the last token are only ever GEMM/element-wise padding: start them at zero (finite); on the
  // stream of the forward pass, which may be a non-blocking stream the null stream does not order with
  PROQA_HIP(hipMemsetAsync(ws.base, 0, total, st));
  char* p = (char*)ws.base;
  _Float16** slots[] = {&ws.h, &ws.h1, &ws.qkv, &ws.ctx, &ws.tmp, &ws.ff, &ws.c_ctx, &ws.c_res, &ws.c_tmp, &ws.c_h1,
                        &ws.c_ff, &ws.c_h, &ws.pooled};
  for (int i = 0; i < 13; ++i) {
    *slots[i] = (_Float16*)p;
    p += sizes[i];
  }
  ws.cu = (int32_t*)p;
  ws.bytes = total;
  ws.rows = rows;
  ws.batch = batch;
  return PROQA_OK;
}

}  // namespace

extern "C" {

int proqa_encoder_create(const proqa_bert_weights* w, proqa_encoder** out) {
  if (!w || !out) return fail(PROQA_EINVAL, "encoder_create: NULL argument");
  *out = nullptr;
  if (w->hidden <= 0 || w->n_heads <= 0 || w->hidden != w->n_heads * 64)
    return fail(PROQA_EINVAL, "encoder_create: hidden=%d must be n_heads*64 (head_dim 64)", w->hidden);
  if (w->n_layers <= 0 || w->intermediate <= 0 || w->intermediate % 8 || w->vocab <= 0 || w->max_position <= 0 || !w->layers)
    return fail(PROQA_EINVAL, "encoder_create: bad model geometry");
  const void* need[] = {w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, w->pool_w, w->pool_b, w->proj_w, w->proj_b};
  for (const void* p : need)
    if (!p) return fail(PROQA_EINVAL, "encoder_create: NULL weight pointer");
  for (int l = 0; l < w->n_layers; ++l) {
    const proqa_bert_layer& L = w->layers[l];
    const void* lp[] = {L.qkv_w, L.qkv_b, L.ao_w, L.ao_b, L.ln1_g, L.ln1_b, L.ff1_w, L.ff1_b, L.ff2_w, L.ff2_b, L.ln2_g, L.ln2_b};
    for (const void* p : lp)
      if (!p) return fail(PROQA_EINVAL, "encoder_create: NULL weight pointer in layer %d", l);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PROQA_ENOGPU, "encoder_create: no HIP device visible");
  proqa_encoder* e = new (std::nothrow) proqa_encoder();
  if (!e) return fail(PROQA_ENOMEM, "encoder_create: out of host memory");
  e->w = *w;
  e->layers.assign(w->layers, w->layers + w->n_layers);
  e->w.layers = e->layers.data();
  PROQA_HIP(hipGetDevice(&e->device));
  if (const char* v = getenv("PROQA_FFN1")) e->own_ffn1 = std::string(v) != "lib";   // developer A/B switch
  rocblas_status s = rocblas_create_handle(&e->blas);
  if (s != rocblas_status_success) {
    delete e;
    return blas_fail(s, "rocblas_create_handle");
  }
  // the large library products on a hipBLASLt kernel pinned by name (lt_gemm.cpp); PROQA_LT_GEMM=0: rocblas_gemm_ex as in rounds 1-3
  if (!(getenv("PROQA_LT_GEMM") && atoi(getenv("PROQA_LT_GEMM")) == 0)) e->lt = lt_gemm_create();
  *out = e;
  return PROQA_OK;
}

int proqa_encoder_set_gemm_tuning(proqa_encoder* e, int enable) {
  if (!e) return fail(PROQA_EINVAL, "encoder_set_gemm_tuning: NULL handle");
  e->tune = enable != 0;
  return PROQA_OK;
}

int proqa_encoder_gemm_kernel(const proqa_encoder* e, char* name_out, size_t name_len) {
  if (!e || !name_out || name_len == 0) return fail(PROQA_EINVAL, "encoder_gemm_kernel: bad argument");
  snprintf(name_out, name_len, "%s", lt_gemm_kernel_name(e->lt));
  return PROQA_OK;
}

int proqa_encoder_workspace(const proqa_encoder* e, void** base_out, size_t* bytes_out) {
  if (!e || !base_out || !bytes_out) return fail(PROQA_EINVAL, "encoder_workspace: NULL argument");
  *base_out = e->ws.base;
  *bytes_out = e->ws.bytes;
  return PROQA_OK;
}

int proqa_encoder_dense(proqa_encoder* e, const void* x, const void* w, void* out, int64_t m, int n, int k, void* stream) {
  if (!e || !x || !w || !out || m < 0 || n <= 0 || k <= 0) return fail(PROQA_EINVAL, "encoder_dense: bad argument");
  PROQA_ON_DEVICE(e->device);
  hipStream_t st = as_stream(stream);
  PROQA_BLAS(rocblas_set_stream(e->blas, st));
  return gemm_tn(e, (const _Float16*)x, w, (_Float16*)out, m, n, k, st);
}

int proqa_encoder_free(proqa_encoder* e) {
  if (!e) return PROQA_OK;
  if (e->ws.base) (void)hipFree(e->ws.base);
  if (e->blas) (void)rocblas_destroy_handle(e->blas);
  lt_gemm_destroy(e->lt);
  delete e;
  return PROQA_OK;
}

int proqa_encoder_forward(proqa_encoder* e, const int64_t* ids_dev, const int32_t* seq_lens_dev, int batch, int seq_len,
                          int64_t n_valid_tokens, int flags, void* out, int out_dtype, void* stream) {
  if (!e || (batch > 0 && (!ids_dev || !seq_lens_dev || !out))) return fail(PROQA_EINVAL, "encoder_forward: NULL argument");
  if (batch < 0 || seq_len <= 0) return fail(PROQA_EINVAL, "encoder_forward: bad sizes");
  if (seq_len > e->w.max_position)
    return fail(PROQA_EINVAL, "encoder_forward: sequence length %d exceeds max_position_embeddings %d", seq_len,
                e->w.max_position);
  if (out_dtype != PROQA_F16 && out_dtype != PROQA_F32) return fail(PROQA_EINVAL, "encoder_forward: bad out dtype");
  if (batch == 0) return PROQA_OK;
  const proqa_bert_weights& w = e->w;
  const int H = w.hidden, I = w.intermediate, NH = w.n_heads;
  const float eps = w.layer_norm_eps;
  const int64_t n_padded = (int64_t)batch * seq_len;
  if (n_valid_tokens > n_padded) return fail(PROQA_EINVAL, "encoder_forward: n_valid_tokens exceeds batch*seq_len");
  // pack only when at least 10 % of the rows are padding: the BLAS default pick is erratic in the row count
  // (scripts/dev_gemm_vs_m.py, up to -15 % per flop on unlucky ragged M), so a nearly full batch is better
  // off at its regular padded shape
  const bool packed = (flags & PROQA_ENC_PACKED) && n_valid_tokens > 0 && n_valid_tokens * 10 < n_padded * 9;
  const bool cls_only = (flags & PROQA_ENC_CLS_ONLY_LAST) != 0;
  // the weights, the workspace and the BLAS handle live on the device the encoder was created on
  PROQA_ON_DEVICE(e->device);   // (restored on return: the caller's current device is not changed under it)
  hipStream_t st = as_stream(stream);
  if (int rc = ensure_workspace(e, batch, round_up<int64_t>(n_padded, kRowTile), st)) return rc;
  Workspace& ws = e->ws;
  PROQA_BLAS(rocblas_set_stream(e->blas, st));

  int64_t n = n_padded;
  const int32_t* cu = nullptr;
  if (packed) {
    n = n_valid_tokens;
    if (int rc = launch_cu_seqlens(seq_lens_dev, batch, ws.cu, stream)) return rc;
    cu = ws.cu;
    if (int rc = proqa_embed_layernorm_varlen_f16(ids_dev, cu, batch, seq_len, H, w.word_emb, w.vocab, w.pos_emb, w.type_emb,
                                                  w.emb_ln_g, w.emb_ln_b, eps, ws.h, stream))
      return rc;
  } else {
    if (int rc = proqa_embed_layernorm_f16(ids_dev, n, seq_len, H, w.word_emb, w.vocab, w.pos_emb, w.type_emb, w.emb_ln_g,
                                           w.emb_ln_b, eps, ws.h, stream))
      return rc;
  }
  // rows handed to the dense layers: the tokens, rounded up to the GEMM tile for large batches (the extra rows
  // hold finite stale values nothing reads back; small batches stay small)
  const int64_t rows = n > 4096 ? round_up<int64_t>(n, kRowTile) : n;
  const int32_t* lens = packed ? nullptr : seq_lens_dev;
  // the fused dense+GELU kernel wants whole 256-row tiles and enough of them to occupy every XCD
  const bool own_ffn1 = e->own_ffn1 && rows % 256 == 0 && rows >= 64 * 256 && I % 256 == 0 && H % 64 == 0;
  _Float16 *h = ws.h, *h1 = ws.h1;
  const int n_full = cls_only ? w.n_layers - 1 : w.n_layers;
  for (int l = 0; l < n_full; ++l) {
    const proqa_bert_layer& L = w.layers[l];
    if (int rc = gemm_tn(e, h, L.qkv_w, ws.qkv, rows, 3 * H, H, st)) return rc;                       // fused Q|K|V projection
    if (int rc = launch_attention(ws.qkv, L.qkv_b, lens, cu, batch, seq_len, NH, ws.ctx, stream)) return rc;
    if (int rc = gemm_tn(e, ws.ctx, L.ao_w, ws.tmp, rows, H, H, st)) return rc;
    if (int rc = proqa_bias_residual_layernorm_f16(ws.tmp, L.ao_b, h, L.ln1_g, L.ln1_b, eps, rows, H, h1, stream)) return rc;
    if (small_dense_ok(rows, I, H)) {
      if (int rc = launch_small_dense(h1, (int)rows, L.ff1_w, L.ff1_b, I, H, 1, ws.ff, stream)) return rc;
    } else if (own_ffn1) {
      // BertIntermediate as ONE launch: the hand-written GEMM adds the bias and applies the erf GELU in its epilogue
      // (gemm_kernels.hip), which saves the 2 x rows x 3072 x 2 B round trip of a separate bias_gelu pass
      if (int rc = proqa_gemm_tn_f16(h1, L.ff1_w, L.ff1_b, ws.ff, rows, I, H, PROQA_GEMM_EPI_BIAS_GELU, stream)) return rc;
    } else {
      if (int rc = gemm_tn(e, h1, L.ff1_w, ws.ff, rows, I, H, st)) return rc;
      if (int rc = proqa_bias_gelu_f16(ws.ff, L.ff1_b, rows, I, stream)) return rc;
    }
    if (int rc = gemm_tn(e, ws.ff, L.ff2_w, ws.tmp, rows, H, I, st)) return rc;
    if (int rc = proqa_bias_residual_layernorm_f16(ws.tmp, L.ff2_b, h1, L.ln2_g, L.ln2_b, eps, rows, H, h, stream)) return rc;
  }
  // h[:, 0] of every sequence -> dst [batch, H]
  auto cls_rows = [&](_Float16* dst) {
    return launch_gather_rows(h, H, cu, seq_len, batch, H, dst, stream);
  };
  if (cls_only) {
    // the pooler reads h[:, 0] only (retriever.py:41-42): the last layer needs K and V of every token but the
    // attention output, both dense blocks and LayerNorms for the [CLS] rows alone
    const proqa_bert_layer& L = w.layers[w.n_layers - 1];
    if (int rc = gemm_tn(e, h, L.qkv_w, ws.qkv, rows, 3 * H, H, st)) return rc;
    if (int rc = launch_attention_cls(ws.qkv, L.qkv_b, lens, cu, batch, seq_len, NH, ws.c_ctx, stream)) return rc;
    if (int rc = cls_rows(ws.c_res)) return rc;
    if (int rc = gemm_tn(e, ws.c_ctx, L.ao_w, ws.c_tmp, batch, H, H, st)) return rc;
    if (int rc = proqa_bias_residual_layernorm_f16(ws.c_tmp, L.ao_b, ws.c_res, L.ln1_g, L.ln1_b, eps, batch, H, ws.c_h1, stream))
      return rc;
    if (small_dense_ok(batch, I, H)) {
      if (int rc = launch_small_dense(ws.c_h1, batch, L.ff1_w, L.ff1_b, I, H, 1, ws.c_ff, stream)) return rc;
    } else {
      if (int rc = gemm_tn(e, ws.c_h1, L.ff1_w, ws.c_ff, batch, I, H, st)) return rc;
      if (int rc = proqa_bias_gelu_f16(ws.c_ff, L.ff1_b, batch, I, stream)) return rc;
    }
    if (int rc = gemm_tn(e, ws.c_ff, L.ff2_w, ws.c_tmp, batch, H, I, st)) return rc;
    if (int rc = proqa_bias_residual_layernorm_f16(ws.c_tmp, L.ff2_b, ws.c_h1, L.ln2_g, L.ln2_b, eps, batch, H, ws.c_h, stream))
      return rc;
  } else {
    if (int rc = cls_rows(ws.c_h)) return rc;
  }
  return proqa_pool_project_f16(ws.c_h, batch, 1, H, w.pool_w, w.pool_b, w.proj_w, w.proj_b, ws.pooled, out, out_dtype, stream);
}

}  // extern "C"

/*
 * proqa_hip.h — C ABI of libproqa_hip.so, the MI355X (gfx950) implementation of
 * ProQA's dense-retrieval hot path: corpus/query encode -> .npy index -> exhaustive
 * inner-product top-k search.
 *
 * The reference (xwhan/ProQA, pure Python) has no FFI of its own: its hot path calls
 * third-party libraries (faiss-cpu, transformers/torch).  Every entry point below
 * names the reference call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PROQA_E* code on failure;
 *     proqa_last_error() gives a thread-local message for the last failure.
 *   - plain pointers and sizes only; no torch / numpy types.  Pointers named *_dev are
 *     device (HBM) addresses valid on the current HIP device; all others are host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *   - the caller allocates every output; the library never frees caller memory.
 *   - handles are not re-entrant: one host thread per handle at a time.
 *   - one process drives one GPU (the current HIP device at handle creation).
 *   - the library has no DT_NEEDED entries for the HIP runtime and rocBLAS (the encoder's dense layers):
 *     the host process loads the pair it already uses with RTLD_GLOBAL before dlopen'ing this library
 *     (INTEGRATION.md section 3), so that device pointers and streams belong to one runtime.
 */
#ifndef PROQA_HIP_H
#define PROQA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PROQA_ABI_VERSION 7

/* element types of embedding matrices (the .npy index is '<f2' under --fp16, else '<f4':
 * retrieval/get_embed.py:139) */
#define PROQA_F16 0
#define PROQA_F32 1

/* error codes */
#define PROQA_OK 0
#define PROQA_EINVAL (-1)   /* bad argument */
#define PROQA_EHIP (-2)     /* HIP runtime error (message carries hipGetErrorString) */
#define PROQA_ENOMEM (-3)   /* allocation failed */
#define PROQA_EIO (-4)      /* file error */
#define PROQA_EFORMAT (-5)  /* malformed .npy */
#define PROQA_ENOGPU (-6)   /* no usable gfx950 device */

/* embedding width is hard-coded to 128 at every consumer in the reference
 * (retrieval/retriever.py:19-20, retrieval/eval_retrieval.py:98) */
#define PROQA_EMBED_DIM 128

const char* proqa_last_error(void);
int proqa_abi_version(void);
/* number of visible HIP devices, and the gcnArchName of the current one (e.g. "gfx950:...") */
int proqa_device_info(int* n_devices, char* arch_name, size_t arch_name_len);

/* ------------------------------------------------------------------------------------
 * Exact inner-product index.  Replaces
 *     index = faiss.IndexFlatIP(d); index.add(xb); D, I = index.search(xq, k)
 * at retrieval/eval_retrieval.py:102-104 (also retrieval/trec_process.py:74-76).
 * Results: scores descending; exact ties ordered by ascending row index (FAISS leaves the
 * tie order unspecified); if fewer than k rows exist the tail is I = -1, D = -FLT_MAX.
 * Scores that are not finite follow faiss's result heap: +inf ranks first; NaN and -inf never compare greater than what
 * the heap holds and are never returned (their slots count as missing rows: I = -1, D = -FLT_MAX at the tail).
 * Rows are stored in HBM as fp16 (the --fp16 index format); scores accumulate in fp32.  float32 data
 * that fp16 cannot hold is searched exactly (see proqa_index_add).
 * ---------------------------------------------------------------------------------- */
typedef struct proqa_index proqa_index;

/* d must be 128.  capacity_rows > 0 preallocates HBM for that many rows (avoids regrowth). */
int proqa_index_create(int d, int64_t capacity_rows, proqa_index** out);
/* append n rows from host memory (row-major [n, d], dtype PROQA_F16 or PROQA_F32).  May be called
 * repeatedly, e.g. once per mmap chunk of para_embed.npy.
 *
 * Precision.  The scan runs on fp16 rows.  float32 input that is exactly representable in fp16 (the
 * float32 upcast of an '<f2' index, retrieval/eval_retrieval.py:99-100) is simply stored as fp16.
 * The first float32 value fp16 cannot hold -- in an added row or in a query -- switches the index to
 * EXACT-FLOAT32 mode: it keeps a float32 copy of every row next to the fp16 roundings; the fp16 scan
 * then only nominates rows (its thresholds are lowered by a rigorous bound on the rounding error,
 * ||q^||.max||x-x^|| + ||q-q^||.max||x^|| + ...), and every nominated row is re-scored from the float32
 * data (double accumulation, rounded once).  Results are the exact float32 top-k; the scan costs the
 * same, the merge more.  Values beyond the fp16 range (|x| > 65504) are refused.
 * proqa_index_allow_rounding(idx, 1) opts out: float32 inputs are rounded to fp16 instead. */
int proqa_index_add(proqa_index* idx, const void* xb, int64_t n, int dtype);
/* np.load(path) + index.add(xb[row0:row0+n]) (retrieval/eval_retrieval.py:99-102) in one call: appends rows
 * [row0, row0 + n) of a 2-D .npy file ('<f2' or '<f4', 128 columns; n < 0 = up to the last row).  n_readers host
 * threads (<= 0: 4) pread() the rows into a ring of four pinned 8 MiB pieces while the pieces read before them travel to
 * HBM, so that the file read, the PCIe transfer and (float32 files) the conversion overlap and no pageable copy of the
 * corpus is made.  A rank of a row-sharded index loads only its own row range this way.  Same precision rules as
 * proqa_index_add.  All or nothing: on failure the index still holds the rows it had. */
int proqa_index_add_npy(proqa_index* idx, const char* path, int64_t row0, int64_t n, int n_readers);
/* same, source already in HBM (used by the synthetic benchmark and the sharded path) */
int proqa_index_add_device(proqa_index* idx, const void* xb_dev, int64_t n, int dtype, void* stream);
/* adopt caller-owned fp16 rows already in HBM without copying; the caller keeps them alive
 * until proqa_index_free/reset.  IMMUTABILITY: searches of k <= 128 over >= 65536 rows scan an int8 copy of the rows
 * (+128 B per row of HBM, see proqa_index_configure_nomination) that the library builds once and keys on the row set; a
 * caller that overwrites adopted rows in place calls proqa_index_rows_changed afterwards (or switches the copy off with
 * proqa_index_configure_nomination(idx, 0)) -- otherwise rows whose new score qualifies are searched against the stale
 * copy and can be missing from the result, silently. */
int proqa_index_adopt_device(proqa_index* idx, const void* xb_dev_f16, int64_t n);
/* the rows of the index were modified in place (adopted memory): whatever the library derived from them (the int8 copy
 * and its statistics) is rebuilt before it is used again.  Not available once the index has switched to exact-float32
 * mode (its float32 copies are the data: reset and add again). */
int proqa_index_rows_changed(proqa_index* idx);
/* Build now what the first search would otherwise build: the int8 copy of the rows for the nomination scan (an
 * allocation of 128 B per row, two passes over the rows, one host read).  Synchronises `stream`.  Optional: a
 * proqa_index_search / _search_device call builds the copy itself when it is missing or stale; an ENQUEUED search
 * (proqa_index_search_begin_device) never allocates or waits -- without a current copy it scans the fp16 rows and its
 * _finish builds the copy -- so a caller that wants its first enqueued search on the int8 copy calls this after add /
 * adopt / rows_changed.  No-op for indexes the nomination scan does not apply to. */
int proqa_index_prepare(proqa_index* idx, void* stream);
/* 1 once the index has switched to exact-float32 mode (see above) */
int proqa_index_is_exact_f32(const proqa_index* idx, int* enabled);
int proqa_index_allow_rounding(proqa_index* idx, int allow);
int proqa_index_ntotal(const proqa_index* idx, int64_t* n);
int proqa_index_reset(proqa_index* idx);
int proqa_index_free(proqa_index* idx);

/* host-pointer search: xq row-major [nq, d]; D float32 [nq, k]; I int64 [nq, k] */
int proqa_index_search(proqa_index* idx, const void* xq, int64_t nq, int dtype, int k,
                       float* D, int64_t* I);
/* device-pointer search; row indices are reported as idx_offset + local row so that a
 * row-sharded corpus yields global ids.  Synchronises `stream` before returning. */
int proqa_index_search_device(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k,
                              int64_t idx_offset, float* D_dev, int64_t* I_dev, void* stream);

/* Two-step form of proqa_index_search_device for a caller that has more work to put on the stream behind the search
 * (the sharded search enqueues its all-gather and merge there).  _begin enqueues the whole search and returns WITHOUT
 * synchronising or allocating whenever it can (k <= 1024 outside the one-pass range, float16 queries or float32 queries
 * of an index already in exact-float32 mode, a workspace that fits -- i.e. not the first search of its size on the
 * handle; other searches run to completion inside _begin).  It never builds the int8 copy of the rows: when that copy is
 * missing or stale the search scans the fp16 rows (same result) and _finish builds the copy after its host wait
 * (proqa_index_prepare builds it ahead of time).  Once the stream reaches that point D_dev / I_dev hold the result, provided no round overflowed its candidate
 * lists (a data-dependent, rare event): status_dev (optional device word, written on the stream) is then 0; 1 means
 * _finish is going to rewrite D_dev / I_dev.  _finish synchronises the stream, runs the overflow-safe re-scan if it is
 * needed (*rewritten = 1; D_dev / I_dev are final when it returns) and completes proqa_index_last_stats.  xq_dev, D_dev,
 * I_dev and status_dev stay valid, and no other call is made on the handle, until _finish has returned.
 * Reference call site: the one faiss search of retrieval/eval_retrieval.py:102-104 (faiss has no asynchronous form). */
int proqa_index_search_begin_device(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset,
                                    float* D_dev, int64_t* I_dev, uint32_t* status_dev, void* stream);
int proqa_index_search_finish(proqa_index* idx, int* rewritten);

/* rows of the index by id, device pointers (faiss reconstruct_batch; the gather `para_embed[I]` of
 * /root/reference/qa/online_sampler.py:117,277 without leaving the GPU): out_dev [n, d] of out_dtype.  PROQA_F16 gives
 * the stored fp16 rows; PROQA_F32 the float32 copies of an exact-float32 index, else exact upcasts of the fp16 rows --
 * in both cases the values add() was given, unless rounding was allowed.  ids are idx_offset + local row as reported
 * by the search; ids outside the index (the -1 of a short result) give zero rows.  Asynchronous on `stream`. */
int proqa_index_reconstruct_batch_device(proqa_index* idx, const int64_t* ids_dev, int64_t n, int64_t idx_offset,
                                         void* out_dev, int out_dtype, void* stream);

/* statistics of the last search on this handle (for tests and the benchmark) */
typedef struct proqa_search_stats {
  int32_t rounds;            /* filter+merge rounds launched */
  int32_t fallback_rounds;   /* rounds run again: slabs re-scanned on the overflow-safe path, and the rounds of the second
                                search that the queries a leaping round left short are given (proqa_index_configure_leap) */
  int64_t candidates;        /* (score,id) pairs that passed the running threshold */
  float filter_ms;           /* HIP-event time of the mips_filter launches, summed over the
                                rounds (0 unless profiling is enabled on the handle) */
  float total_ms;            /* HIP-event time of the whole search on the stream */
  int64_t nominated;         /* rows the int8 nomination rounds handed to the exact re-scoring (0 for an fp16 scan) */
  int32_t nomination;        /* 1 if the rounds of this search scanned the int8 copy of the rows, else 0 */
  int32_t nomination_state;  /* of the index after this search: 0 off (mode 0 / exact-float32), 1 on (the copy is, or will
                                be, scanned by the searches it applies to), 2 suspended (the rows do not quantise, or a
                                search over-nominated: fp16 scan until a re-probe succeeds or the rows change) */
  int32_t leap_rank;         /* > 0: the rounds of this search tested against the score at that rank (< k) of the running lists
                                (leaping rounds, proqa_index_configure_leap); 0: against the k-th best, as ever */
  int32_t leap_state;        /* of the index after this search: 0 off (mode 0 / exact-float32), 1 on, 2 paused (leaping
                                rounds fell short: ordinary rounds for the next 16, 32, ... 1024 eligible searches) */
} proqa_search_stats;
int proqa_index_last_stats(const proqa_index* idx, proqa_search_stats* out);
/* bracket every mips_filter launch with HIP events on the search stream (bench.py roofline) */
int proqa_index_set_profiling(proqa_index* idx, int enable);
/* tuning knobs of the round schedule (0 keeps the default): rows of the first slab (scanned with
 * threshold -inf) and the growth factor of the following slabs */
int proqa_index_configure(proqa_index* idx, int first_slab_rows, int growth);
/* rows whose exact top-k a search takes from a dense score matrix (two launches) before the threshold rounds start;
 * a multiple of 32 up to 8192, 0 disables it.  Used for k <= rows/4 (at most 256) on indexes of >= 4*rows rows;
 * the result never depends on it.  Default 4096. */
int proqa_index_configure_bootstrap(proqa_index* idx, int rows);
/* The int8 nomination scan of the k <= 128 rounds (fp16 indexes of >= 65536 rows): the
 * rounds scan an int8 copy of the centred, per-dimension-scaled rows at twice the fp16 MFMA rate, nominate every row whose
 * integer score exceeds the running threshold lowered by a rigorous bound on the quantisation error, and re-score the
 * nominated rows from the fp16 rows -- the result is the fp16 scan's, bit for bit.  mode 0: never (fp16 scan);
 * 1 (default, automatic): a search that re-scores more than max(4096, rows / 2048) rows per query or needs the
 * overflow-safe path SUSPENDS the int8 rounds (fp16 scan); the 8th eligible search after that probes them again, a
 * failed probe doubles the distance (16, 32, 64, 64, ...), a successful one resumes; a change of the rows starts afresh;
 * every switch is one line on stderr when PROQA_LOG is set, and proqa_search_stats.nomination_state reports the state;
 * 2: always.  A search with k in the thousands for a FEW queries (one question, k = 5000: rows >= 16 x queries x ~1.5 k) scans
 * the copy as well, in its one launch over the shard (same result; mode 0 keeps the fp16 rows).  The copy (+128 B per row) is
 * built by proqa_index_prepare, or by the first search that wants it.  Rows
 * adopted with proqa_index_adopt_device: see the immutability note there. */
int proqa_index_configure_nomination(proqa_index* idx, int mode);
/* Leaping rounds of the k <= 128 searches (fp16 indexes, default schedule, behind the bootstrap).  A round that tests against
 * the k-th best score of the n0 rows seen so far logs ~k (rho - 1) candidates per query while the rows seen grow by rho --
 * which is what keeps rho near 3 and an 18M-row search at eight rounds.  A LEAPING round tests against the score at rank
 * j < k of the running list: it logs ~j (rho - 1) rows, so rho can be 5-16 and the rounds three or four.  Its result is exact
 * iff at least k rows of everything seen so far beat that threshold (every one of them is then known); the round's merge
 * checks exactly that, and a round that falls short is re-scanned against the k-th best scores on the overflow-safe path
 * (proqa_search_stats.fallback_rounds) -- the result never depends on the leap.  j is the smallest rank for which the
 * shortfall has probability <= 1e-8 per query and round when the first rows stand for the rest (rows in no particular
 * order: the count of new rows above the rank-j score is negative-binomial (j, 1 / rho)).  What a shortfall costs: the
 * queries that fell short are searched again as a batch of their own on ordinary rounds and their result rows rewritten
 * (every other query's result is verified): ~ +10 % of the search for a handful, at most one more search for all of them
 * (the flagged slabs are re-scanned for all queries only when a list or a merge overflowed as well).  Rows sorted by topic
 * or norm -- or long runs of near-duplicates: ~k/2 rows that score alike next to each other -- make leaps fall short:
 * up to 256 short queries count three strikes (a clean leap takes one back), more eight; at eight strikes the leaps of the
 * index pause for 16
 * eligible searches, a failed retry doubles the pause (up to 1024), a clean one clears it, changed rows start afresh; one
 * stderr line per event under PROQA_LOG.  mode 0: never; 1 (default): automatic, as above.  An index configured with
 * proqa_index_configure(growth) keeps its ordinary rounds. */
int proqa_index_configure_leap(proqa_index* idx, int mode);
/* The schedule proqa_index_configure_leap's automatic mode takes for a search of `queries` queries for the best k of `rows`
 * rows behind a bootstrap of `bootstrap_rows` (host arithmetic only, no GPU): rounds behind the bootstrap (0: no leap --
 * k > 128 or k < 8, or nothing fits the merges), the rank j < k its thresholds sit at, the rows expected above a threshold
 * per round and query (j (rho - 1), rho = (rows / bootstrap_rows)^(1 / rounds)) and the probability that fewer than k - j rows
 * beat it -- the negative-binomial (j, 1 / rho) sum -- per query and round.  nominating: the rounds scan the int8 copy (their
 * merges hold 2048 nominated rows).  rows_per_round / shortfall_probability may be NULL. */
int proqa_leap_plan(int64_t rows, int64_t bootstrap_rows, int k, int64_t queries, int nominating, int* rounds, int* rank,
                    double* rows_per_round, double* shortfall_probability);

/* Merge n_parts per-shard result lists into one: D_parts/I_parts are [n_parts, nq, k]
 * (the layout an RCCL all-gather of per-rank [nq, k] produces); every list as a search reports it (scores descending,
 * ties by ascending id, the I = -1 slots of a short shard at its tail), parts in ascending order of their row ranges.
 * Same ordering rule.  These two entry points SORT what they are given (no assumption about the order inside a part: a
 * caller's own local search may have produced it): up to 16384 gathered keys per query in LDS, larger merges (k = 10000 x
 * 8 shards) as a segmented radix sort in HBM (temporary buffers are allocated for the call).  The sharded search's own
 * merge (proqa_topk_merge_gathered_device, below) knows its parts are search results and merges up to 4096 keys per query
 * (8 shards x top-80) by rank, without a sort. */
int proqa_topk_merge_device(const float* D_parts_dev, const int64_t* I_parts_dev, int n_parts,
                            int64_t nq, int k, float* D_dev, int64_t* I_dev, void* stream);
/* The same merge over parts that are not back to back: part p's [nq, k] scores / ids start stride_d / stride_i ELEMENTS
 * after part p-1's -- the receive buffer of ONE all-gather whose per-rank block holds the ids followed by the scores
 * is merged where it lies (D_parts = buffer + ids bytes, stride_d = block bytes / 4, stride_i = block bytes / 8). */
int proqa_topk_merge_strided_device(const float* D_parts_dev, const int64_t* I_parts_dev, int n_parts, int64_t nq, int k,
                                    int64_t stride_d, int64_t stride_i, float* D_dev, int64_t* I_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Encoder: BertForRetriever.get_embed (retrieval/retriever.py:33-43 + transformers BertModel)
 * as one call: embeddings -> n_layers x {QKV, attention, output dense + LN, FFN + LN} ->
 * pooler (tanh dense on h[:,0]) -> Linear(hidden, 128).
 * All weights are device pointers to fp16, dense weights row-major [out, in] (the layout of the
 * reference checkpoint, cast to fp16); qkv_w / qkv_b are the query|key|value weights stacked
 * along `out` ([3*hidden, hidden] / [3*hidden]).  The caller owns the weights and keeps them
 * alive while the encoder exists.  The dense layers are rocBLAS GEMMs (fp32 accumulate).
 * ------------------------------------------------------------------------------------ */
typedef struct proqa_bert_layer {
  const void *qkv_w, *qkv_b;      /* [3*hidden, hidden], [3*hidden] */
  const void *ao_w, *ao_b;        /* attention.output.dense */
  const void *ln1_g, *ln1_b;      /* attention.output.LayerNorm */
  const void *ff1_w, *ff1_b;      /* intermediate.dense  [intermediate, hidden] */
  const void *ff2_w, *ff2_b;      /* output.dense        [hidden, intermediate] */
  const void *ln2_g, *ln2_b;      /* output.LayerNorm */
} proqa_bert_layer;

typedef struct proqa_bert_weights {
  int32_t hidden, n_layers, n_heads, intermediate, max_position;
  int64_t vocab;
  float layer_norm_eps;
  const void *word_emb, *pos_emb, *type_emb;   /* type_emb: row 0 (token_type_ids are never passed) */
  const void *emb_ln_g, *emb_ln_b;
  const proqa_bert_layer* layers;              /* n_layers entries (copied by create) */
  const void *pool_w, *pool_b;                 /* pooler.dense */
  const void *proj_w, *proj_b;                 /* proj_q / proj_c: [128, hidden], [128] */
} proqa_bert_weights;

typedef struct proqa_encoder proqa_encoder;

#define PROQA_ENC_CLS_ONLY_LAST 1  /* last layer: attention output / dense blocks / LayerNorms for the [CLS] rows only */
#define PROQA_ENC_PACKED 2         /* evaluate the valid tokens only (needs n_valid_tokens) */

int proqa_encoder_create(const proqa_bert_weights* w, proqa_encoder** out);
int proqa_encoder_free(proqa_encoder* enc);
/* opt-in (off by default): the first forward that meets a large GEMM shape times every rocBLAS solution for
 * it on the real operands (~0.3 s per shape) and keeps the winner if an interleaved re-match confirms >= 2 %;
 * same arithmetic (fp16 in, fp32 accumulate), possibly another summation order. */
int proqa_encoder_set_gemm_tuning(proqa_encoder* enc, int enable);
/* Which library kernel runs the large dense layers (QKV, attention output, FFN2; FFN1 is proqa_gemm_tn_f16).  hipBLASLt's
 * own choice for these shapes is 3-20 % slower than the best kernel it holds, so the encoder calls hipBLASLt itself with an
 * algorithm chosen BY KERNEL NAME from a fixed preference list (csrc/lt_gemm.cpp) -- deterministic, no timing at run time;
 * with a library build that does not know the names, or without hipBLASLt in the process, the layers run on
 * rocblas_gemm_ex.  name_out receives the pinned kernel's name ("" = rocblas_gemm_ex: no name matched, or no large product
 * has run yet on this handle), truncated to name_len - 1 characters. */
int proqa_encoder_gemm_kernel(const proqa_encoder* enc, char* name_out, size_t name_len);
/* The activation workspace of the handle: base address and size (0 / 0 before the first forward).  A forward over more
 * tokens or sequences than any before it REPLACES the workspace; a caller that captured forwards of this handle into a HIP
 * graph (proqa_amd.online_retriever.GraphedQuestionEncoder) compares the base address before a replay and re-captures when
 * it has changed.  forward itself is capturable once the workspace fits: no allocation, no host synchronisation. */
int proqa_encoder_workspace(const proqa_encoder* enc, void** base_out, size_t* bytes_out);
/* The library dense layer of this handle on its own (tests): out[m, n] = x[m, k] . w[n, k]^T, fp16 row-major device
 * pointers, fp32 accumulate -- the pinned hipBLASLt kernel where it applies, else rocblas_gemm_ex (small_dense_mfma for
 * <= 256 rows).  Asynchronous on `stream`. */
int proqa_encoder_dense(proqa_encoder* enc, const void* x_dev, const void* w_dev, void* out_dev, int64_t m, int n, int k,
                        void* stream);
/* ids_dev: [batch, seq_len] int64, right-padded (retrieval/datasets.py:29-45); seq_lens_dev: [batch]
 * int32 valid lengths (>= 1); n_valid_tokens: their sum if the host knows it, else -1 (then the
 * padded layout is evaluated whatever the flags say); out: [batch, 128] of out_dtype.
 * Asynchronous on `stream`; not re-entrant per handle. */
int proqa_encoder_forward(proqa_encoder* enc, const int64_t* ids_dev, const int32_t* seq_lens_dev, int batch,
                          int seq_len, int64_t n_valid_tokens, int flags, void* out, int out_dtype, void* stream);

/* The kernels the encoder is made of, individually (tests, other drivers). */
/* y[m,n] = epilogue(x[m,k] . w[n,k]^T + bias[n]): the encoder's dense layer as a hand-written MFMA GEMM (fp16 in, fp32
 * accumulate, fp16 out; 256 x 256 x 64 tiles, LDS-DMA ring).  epilogue 0: none (bias ignored), 1: + bias,
 * 2: erf-GELU(. + bias) = BertIntermediate.  m and n multiples of 256, k a multiple of 64. */
#define PROQA_GEMM_EPI_NONE 0
#define PROQA_GEMM_EPI_BIAS 1
#define PROQA_GEMM_EPI_BIAS_GELU 2
int proqa_gemm_tn_f16(const void* x_dev, const void* w_dev, const void* bias_dev, void* y_dev, int64_t m, int n, int k,
                      int epilogue, void* stream);
/* out[b,s,:] = LayerNorm(word[ids[b,s]] + pos[s] + type[0]) , eps = 1e-12
 * (BertEmbeddings; token_type_ids are never passed by the reference => row 0) */
int proqa_embed_layernorm_f16(const int64_t* ids_dev, int64_t n_tokens, int seq_len, int hidden,
                              const void* word_emb, int64_t vocab, const void* pos_emb,
                              const void* type_emb, const void* ln_gamma, const void* ln_beta,
                              float eps, void* out, void* stream);

/* Packed ("varlen") token layout: sequences are stored back to back without padding rows,
 * cu_seqlens[b] = first row of sequence b, cu_seqlens[batch] = total tokens (int32, device).
 * Every per-token operator is unchanged on the packed [T, hidden] matrix; only the embedding
 * gather and the attention need the offsets.  Results for the valid tokens equal the padded
 * evaluation (padding never reaches a valid token: retrieval/datasets.py:29-45 pads on the right,
 * the attention masks those keys).
 * ids stay padded [batch, seq_len] (as em_collate produces them); out_packed is [T, hidden]. */
int proqa_embed_layernorm_varlen_f16(const int64_t* ids_dev, const int32_t* cu_seqlens_dev, int batch,
                                     int seq_len, int hidden, const void* word_emb, int64_t vocab,
                                     const void* pos_emb, const void* type_emb, const void* ln_gamma,
                                     const void* ln_beta, float eps, void* out_packed, void* stream);

/* fused multi-head self-attention for one layer:
 *   ctx = softmax(Q K^T / sqrt(64) + key_mask) V,   head_dim 64
 * qkv is the fused projection output [B*S, 3*hidden] (Q | K | V, each head-major);
 * seq_lens[b] = number of valid (unpadded) keys of sequence b — the reference pads on the
 * right with mask False (retrieval/datasets.py:29-45,298-305). */
int proqa_attention_f16(const void* qkv, const int32_t* seq_lens_dev, int batch, int seq_len,
                        int n_heads, void* ctx_out, void* stream);

/* the same attention for query row 0 ([CLS]) of every sequence only: ctx_cls_out [B, hidden].
 * Only h[:, 0] of the last layer reaches the pooler (retrieval/retriever.py:41-42), so the last
 * layer's attention output / FFN are computed for that row alone. */
int proqa_attention_cls_f16(const void* qkv, const int32_t* seq_lens_dev, int batch, int seq_len,
                            int n_heads, void* ctx_cls_out, void* stream);

/* the two attention entry points on the packed layout: qkv_packed [T, 3*hidden], ctx_packed_out
 * [T, hidden] (ctx_cls_out stays [batch, hidden]); max_seq_len = longest sequence of the batch */
int proqa_attention_varlen_f16(const void* qkv_packed, const int32_t* cu_seqlens_dev, int batch,
                               int max_seq_len, int n_heads, void* ctx_packed_out, void* stream);
int proqa_attention_cls_varlen_f16(const void* qkv_packed, const int32_t* cu_seqlens_dev, int batch,
                                   int max_seq_len, int n_heads, void* ctx_cls_out, void* stream);
/* general form of the four entry points above, with the bias of the fused Q|K|V projection folded in
 * (qkv_bias [3*hidden] fp16 or NULL): the query bias is added to the query fragments, the key bias is
 * dropped (it shifts all scores of a query by the same amount: softmax-invariant), the value bias is
 * added to the output (probabilities sum to 1).  Exactly one of seq_lens_dev (padded layout) and
 * cu_seqlens_dev (packed layout) is non-NULL; cls_only selects the [batch, hidden] row-0 output. */
int proqa_attention_ex_f16(const void* qkv, const void* qkv_bias, const int32_t* seq_lens_dev,
                           const int32_t* cu_seqlens_dev, int batch, int seq_len, int n_heads, int cls_only,
                           void* out, void* stream);

/* x = gelu_erf(x + bias) in place, x [rows, cols] (BertIntermediate, hidden_act='gelu') */
int proqa_bias_gelu_f16(void* x, const void* bias, int64_t rows, int cols, void* stream);

/* out = LayerNorm(x + bias + residual) (BertSelfOutput / BertOutput), eps = 1e-12 */
int proqa_bias_residual_layernorm_f16(const void* x, const void* bias, const void* residual,
                                      const void* gamma, const void* beta, float eps,
                                      int64_t rows, int cols, void* out, void* stream);

/* embed[b,:] = (tanh(h[b,0,:] Wp^T + bp)) Wproj^T + bproj
 * (BertPooler + proj_{q,c}: retrieval/retriever.py:19-20,37-42).  h is [B, S, hidden];
 * pooled_ws is caller-provided scratch of B*hidden fp16; out is [B, 128] in out_dtype. */
int proqa_pool_project_f16(const void* h, int batch, int seq_len, int hidden, const void* w_pool,
                           const void* b_pool, const void* w_proj, const void* b_proj,
                           void* pooled_ws, void* out, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * k-means over passage embeddings.  Replaces faiss.Clustering.train + index.search(data, 1) of
 * retrieval/group_paras.py:20-53 (IndexFlatL2, IndexFlatIP when --spherical).  The Lloyd loop
 * (sampling, initialisation, empty-cluster splitting: faiss Clustering.cpp) runs on the host
 * (proqa_amd/group_paras.py); these are its two heavy steps.  Points are fp16 [n,128] in HBM,
 * centroids float32 [k,128] in HBM (fp32-grade dot products: hi/lo split on the matrix pipe).
 * ---------------------------------------------------------------------------------- */
typedef struct proqa_kmeans proqa_kmeans;
int proqa_kmeans_create(int d, int64_t n_max, int k, proqa_kmeans** out);
int proqa_kmeans_free(proqa_kmeans* h);
/* nearest centroid of every point: assign int32 [n]; dist float32 [n] = squared L2 distance
 * (metric_l2 != 0) or inner product; exact ties go to the lowest centroid index */
int proqa_kmeans_assign_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const float* centroids_dev,
                               int metric_l2, int32_t* assign_dev, float* dist_dev, void* stream);
/* the same with a hint per point (int32 [n], may be NULL, may alias assign_dev): any centroid index -- a Lloyd loop passes
 * the assignment of its previous iteration.  The result does not depend on the hint; a good one lets the nominating pass
 * skip most of its bookkeeping (a hint out of [0, k) is ignored for that point). */
int proqa_kmeans_assign_hinted_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const float* centroids_dev,
                                      int metric_l2, const int32_t* hint_dev, int32_t* assign_dev, float* dist_dev,
                                      void* stream);
/* centroid c = mean of its points, summed in fp32 in point order (faiss km_update_centroids);
 * counts uint32 [k]; centroids of empty clusters are left untouched */
int proqa_kmeans_update_device(proqa_kmeans* h, const void* x_f16_dev, int64_t n, const int32_t* assign_dev,
                               float* centroids_dev, uint32_t* counts_dev, void* stream);
/* faiss' rand_perm(perm, n, seed) (std::mt19937, i2 = i + rng() % (n - i)): the permutation faiss
 * uses to sub-sample the training set and to pick the initial centroids */
int proqa_rand_perm(int64_t n, int64_t seed, int32_t* perm_out);

/* ------------------------------------------------------------------------------------
 * .npy index files.  Replace np.save (retrieval/get_embed.py:139) and np.load
 * (retrieval/eval_retrieval.py:99-100) for 2-D C-order '<f2' / '<f4' arrays, format v1.0,
 * header padded so that data starts at a multiple of 64 bytes.
 * ---------------------------------------------------------------------------------- */
typedef struct proqa_npy_info {
  int64_t rows;
  int64_t cols;
  int32_t dtype;        /* PROQA_F16 / PROQA_F32 */
  int64_t data_offset;  /* byte offset of element [0,0] */
} proqa_npy_info;

int proqa_npy_stat(const char* path, proqa_npy_info* info);
/* read rows [row0, row0+n) into dst (host), element type as stored */
int proqa_npy_read_rows(const char* path, int64_t row0, int64_t n, void* dst, size_t dst_bytes);
/* write a whole array */
int proqa_npy_write(const char* path, const void* data, int64_t rows, int64_t cols, int dtype);
/* create a pre-sized file (header + zero-filled data) that ranks later fill by row range */
int proqa_npy_create(const char* path, int64_t rows, int64_t cols, int dtype);
/* src: n rows of `cols` elements of `dtype`; both must match the file's header (a rank writing float32 rows into a
 * '<f2' file is refused) */
int proqa_npy_write_rows(const char* path, int64_t row0, int64_t n, const void* src, int64_t cols, int dtype);

/* ------------------------------------------------------------------------------------
 * WordPiece tokenisation on the host, native and multi-threaded: `tokenizer.encode(sent, max_length=L)` of
 * retrieval/datasets.py:285-286 (transformers' BertTokenizer: clean-up, CJK spacing, NFD + accent stripping + lower-casing
 * for uncased models, punctuation splitting, greedy WordPiece) for every text of the Basic Multilingual Plane without a
 * '[', table-driven (csrc/wordpiece_tables.inc, generated from the tokenizers library), so that get_embed.py's loader does
 * not depend on ~3.5 k passages/s per thread of Python-bound tokenizer calls.  The few other texts are flagged and left to
 * the caller's reference tokenizer.
 * ---------------------------------------------------------------------------------- */
typedef struct proqa_wordpiece proqa_wordpiece;
/* vocab: the tokens of vocab.txt joined by '\n' (token i has id i), vocab_bytes long; do_lower_case: bit 0 as the model's,
 * bit 1 = ASCII only (every text with a byte >= 0x80 is flagged -1: for callers whose reference tokenizer is NOT the
 * tokenizers library the tables were generated from -- the pure-Python BertTokenizer lower-cases whole strings (final sigma),
 * NFC-normalises first and carries its own Unicode data version) */
int proqa_wordpiece_create(const char* vocab, size_t vocab_bytes, int do_lower_case, proqa_wordpiece** out);
int proqa_wordpiece_free(proqa_wordpiece* tok);
/* texts[i] (text_bytes[i] bytes, UTF-8, not NUL-terminated) -> ids_out[i, 0..max_length) = [CLS] pieces [SEP] truncated
 * to max_length, zero-padded; lens_out[i] = its length, or -1 if text i is left to the caller's reference tokenizer (a '[',
 * a character beyond the Basic Multilingual Plane, one of 15 reordering-sensitive marks, malformed UTF-8; row i of ids_out
 * is then unspecified).  n_threads host threads share the batch. */
int proqa_wordpiece_encode_batch(const proqa_wordpiece* tok, const char* const* texts, const int64_t* text_bytes, int64_t n,
                                 int max_length, int64_t* ids_out, int32_t* lens_out, int n_threads);
/* The same straight from the records of the JSON-lines input file (retrieval/datasets.py:271-283: json.loads(line) and
 * sample['text'] / sample['question']): lines[i] (line_bytes[i] bytes, surrounding white space allowed) is one JSON object,
 * the string value of its top-level member `key` (NUL-terminated) is tokenised.  lens_out[i] = -2 for a record this parser
 * does not take -- anything but a flat object of strings, numbers and true / false / null, a surrogate escape, a missing or
 * non-string member, any deviation from the JSON grammar: the caller's json module reads (or rejects) those lines. */
int proqa_wordpiece_encode_jsonl_batch(const proqa_wordpiece* tok, const char* const* lines, const int64_t* line_bytes,
                                       int64_t n, const char* key, int max_length, int64_t* ids_out, int32_t* lens_out,
                                       int n_threads);

/* ------------------------------------------------------------------------------------
 * Multi-GPU search without PyTorch (BASELINE.json configs[3]).  One process (or thread)
 * per GPU holds rows [lo, hi) of the corpus in its own proqa_index and all queries.  A sharded search is the
 * local exact top-k with global ids, ONE RCCL all-gather of the per-rank [nq, k] (id, score) lists over xGMI,
 * and the merge of proqa_topk_merge_device: bit-identical to searching the whole corpus on one GPU.
 * The reference has no such call (its search is single-process faiss, retrieval/eval_retrieval.py:102-104; its
 * one NCCL touch point is the process-group init of retrieval/get_embed.py:44-52).
 * RCCL is bound at run time (dlopen); in a PyTorch process the RCCL PyTorch loaded is used.
 * ---------------------------------------------------------------------------------- */
#define PROQA_COMM_ID_BYTES 128
typedef struct proqa_comm proqa_comm;
/* rank 0: make a unique id (ncclGetUniqueId) and hand its 128 bytes to every rank by any side channel */
int proqa_comm_get_unique_id(void* id_out);
/* every rank, with its GPU current (hipSetDevice): collective ncclCommInitRank */
int proqa_comm_create(const void* id, int world_size, int rank, proqa_comm** out);
int proqa_comm_info(const proqa_comm* comm, int* world_size, int* rank);
int proqa_comm_free(proqa_comm* comm);
/* every rank calls with the SAME queries (device pointers), nq, dtype and k, and with idx_offset = its first global
 * row; ranks must hold ascending row ranges in rank order.  D_dev / I_dev [nq, k] receive the merged result on every
 * rank.  The collective runs even for world_size 1.  The local search, the all-gather and the merge are enqueued back to
 * back (proqa_index_search_begin_device) and the host waits once, at the end; each rank's block carries a status word, so
 * that a rank whose lists overflowed triggers ONE more exchange on every rank, and a rank whose local search fails still
 * enters the collective: every rank then returns an error instead of waiting for it forever.  One stream per communicator
 * at a time (the exchange buffers belong to the communicator). */
int proqa_sharded_search_device(proqa_index* idx, proqa_comm* comm, const void* xq_dev, int64_t nq, int dtype,
                                int k, int64_t idx_offset, float* D_dev, int64_t* I_dev, void* stream);
/* The pieces of that call for a caller that brings its own collective (proqa_amd.index.ShardedIndexFlatIP over
 * torch.distributed): the per-rank block the all-gather exchanges is [ids int64 nq*k | scores float nq*k | status word],
 * every part padded to 16 bytes (_block_layout); the rank-local search writes ids, scores and the status word straight
 * into its block (proqa_index_search_begin_device), and _merge_gathered merges the n_parts blocks of the receive buffer
 * where they lie and drops every block's status word into status_host (n_parts words of PINNED host memory, e.g.
 * hipHostMalloc; read them after synchronising the stream; NULL skips it): 0 = that rank's list stands, 1 = it is being
 * rewritten by proqa_index_search_finish (exchange once more), 0xFFFFFFFF = that rank failed. */
int proqa_sharded_block_layout(int64_t nq, int k, size_t* ids_bytes, size_t* scores_bytes, size_t* block_bytes);
int proqa_topk_merge_gathered_device(const void* gathered_dev, int n_parts, int64_t nq, int k, uint32_t* status_host,
                                     float* D_dev, int64_t* I_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Measured ceilings of this GPU (bench.py "peak_measured"; SURVEY.md section 8d asks for them next to the spec
 * peaks): a 16-byte-per-lane stream over device memory and a register-resident 32x32x16 fp16 MFMA loop.
 * ---------------------------------------------------------------------------------- */
/* kind 0: copy buf[0,bytes) -> buf[bytes,2*bytes), read+written bytes counted; kind 1: read buf[0,bytes).
 * Best of `reps` launches in GB/s. */
int proqa_microbench_stream(void* buf_dev, size_t bytes, int kind, int reps, void* stream, double* gbs);
/* launches of about ms_target milliseconds; zero_operands != 0 feeds all-zero inputs (the sustained clock depends on
 * the operand bits).  Dense TFLOP/s. */
int proqa_microbench_mfma(double ms_target, int zero_operands, void* stream, double* tflops);
/* the same loop on v_mfma_i32_32x32x32_i8 (the nomination scan's instruction); dense TOP/s */
int proqa_microbench_mfma_i8(double ms_target, int zero_operands, void* stream, double* tops);
/* the same with the instruction shape as an argument: 0 = v_mfma_i32_32x32x32_i8 (the nomination scan's), 1 =
 * v_mfma_i32_16x16x64_i8 (the shape MI355X_MICROARCH.md quotes its int8 ceiling for) */
int proqa_microbench_mfma_i8_shape(double ms_target, int zero_operands, int shape, void* stream, double* tops);
/* what VALU work beside the matrix instructions costs: the int8 rate (random operands) of a loop of 8 x
 * v_mfma_i32_32x32x32_i8 + n_valu x v_xad_u32 per wave and trip, four waves per SIMD (the shape of mips_filter_i8's unit);
 * n_valu in {0, 8, 16, 24, 32, 48, 64} */
int proqa_microbench_mfma_i8_valu(double ms_target, int n_valu, void* stream, double* tops);
/* microseconds of ONE cooperative launch of `grid` 256-thread workgroups that meet at n_syncs grid-wide barriers
 * (hipLaunchCooperativeKernel + cooperative_groups::grid_group::sync): what fusing dependent small kernels would pay */
int proqa_microbench_grid_sync(int grid, int n_syncs, void* stream, double* us);

#ifdef __cplusplus
}
#endif
#endif /* PROQA_HIP_H */

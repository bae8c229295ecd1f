// Host orchestration of the exact inner-product index (proqa_index_* in proqa_hip.h).
//
// Replaces faiss.IndexFlatIP at /root/reference/retrieval/eval_retrieval.py:102-104:
//   add()    -> rows live in HBM as fp16 [N,128] (the --fp16 .npy format, get_embed.py:139)
//   search() -> rounds of {mips_filter_f16 over a corpus slab, topk_merge per query}; slabs grow
//               geometrically so the per-query threshold (running k-th best score) tightens
//               fast and the big late slabs emit only a handful of candidates per query.
//   If a round overflows a query's candidate list (adversarially ordered corpus), its slab is
//   re-scanned on the overflow-safe path: sub-slabs of <= capacity rows, inclusive threshold,
//   duplicate removal in the merge.  The result is exact either way.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>
#include <cerrno>

#include "common.h"
#include "mips_kernels.h"

namespace proqa {

struct Slab {
  long long r0, r1;
};

}  // namespace proqa

struct proqa_index {
  int device = 0;
  // corpus
  char* xb = nullptr;       // fp16 rows (in exact-float32 mode: the fp16 roundings the filter scans)
  bool owns_xb = true;
  // exact-float32 mode: float32 inputs fp16 cannot hold keep a float32 copy of every row; the fp16
  // filter then only nominates rows (threshold lowered by a rigorous error margin) and the merge
  // re-scores them from xb32
  bool exact = false;
  float* xb32 = nullptr;
  int64_t capacity32 = 0;
  int64_t rows32 = 0;                      // float32 rows written so far (>= n while an add call is uploading its pieces)
  unsigned* norm_stats = nullptr;          // device: {max ||x - fp16(x)||, max ||fp16(x)||} as float bits
  float* xq32 = nullptr;                   // workspace [ws_nq_pad,128]
  float* margin = nullptr;                 // workspace [ws_nq_pad]
  float* tau_filter = nullptr;             // workspace [ws_nq_pad]
  float* ub_filter = nullptr;              // workspace [ws_nq_pad]
  int64_t n = 0;
  int64_t capacity = 0;
  // search workspace (grown on demand)
  int64_t ws_nq_pad = 0;
  int ws_k = 0;
  void* xq_pad = nullptr;
  float* tau = nullptr;
  unsigned* run_n = nullptr;
  unsigned long long* run_keys = nullptr;
  // paged search (k > kPageK): per query bound of the next page
  unsigned long long* bound_keys = nullptr;
  float* ub = nullptr;
  unsigned char* done = nullptr;
  // candidate records of one filter launch (see CandidateStore in mips_kernels.h)
  proqa::WaveRecord* lane_log = nullptr;
  unsigned* lane_cnt = nullptr;
  proqa::WaveRecord* spill_log = nullptr;
  unsigned* spill_cnt = nullptr;
  size_t store_records = 0;                // what the store holds: lane-list records, lane lists, spill slots
  size_t store_lists = 0;
  size_t store_slots = 0;
  unsigned* overflow = nullptr;            // [kMaxRounds] device
  proqa::SearchMirror* mirror = nullptr;   // pinned host memory the finalize kernel reports into (overflow words, candidates)
  unsigned long long* stat_dev = nullptr;  // [ws_nq_pad] candidates per query (part of the workspace)
  void* stage_dev = nullptr;               // staging for host-pointer add/search
  size_t stage_bytes = 0;
  void* stage_pinned = nullptr;
  size_t stage_pinned_bytes = 0;
  hipStream_t io_stream = nullptr;         // stream of the host-pointer search (proqa_index_search)
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipEvent_t ev_filter[2 * 96] = {};       // per-round brackets, created when profiling is on
  bool profile = false;
  bool allow_rounding = false;             // accept fp32 inputs that are not exactly representable in fp16
  unsigned* inexact = nullptr;             // device counters: {not exact in fp16, beyond the fp16 range}
  // bootstrap (see mips_kernels.hip): score matrix of the first rows, [ws_nq_pad, boot_stride] floats
  float* boot_scores = nullptr;
  size_t boot_floats = 0;
  int bootstrap_rows = 4096;               // 0 disables it
  bool bootstrap_auto = true;              // not configured by the caller: 8192 rows where that saves candidates (page_enqueue)
  // int8 nomination scan (mips_kernels.hip): an int8 copy of the centred, per-dimension-scaled rows, built lazily by the first
  // search that can use it and rebuilt when the rows changed (rows_epoch); the fp16 rows stay the data every score comes from
  signed char* xb8 = nullptr;
  float2* blk8 = nullptr;                  // per 32-row block: scale and largest integer row norm (FilterArgsI8::blk)
  int64_t capacity8 = 0;
  float* col = nullptr;                    // device [3][128]: mean, 127 / c, c / 127
  float* col_partial = nullptr;            // device [kColStatGroups][3][128]
  proqa::QuantStats* qstats = nullptr;     // device
  uint64_t rows_epoch = 1;                 // bumped by every change of the rows
  uint64_t q8_epoch = 0;                   // rows_epoch the int8 copy was built for (0: never)
  bool q8_usable = false;                  // that copy can be scanned (finite statistics)
  bool q8_unprofitable = false;            // a search on this copy nominated far too many rows: the int8 rounds are SUSPENDED (fp16 scan)
  // ... until a re-probe succeeds: the q8_probe_after-th eligible search after the suspension runs the int8 rounds again;
  // a probe that fails doubles the distance (8, 16, 32, 64, 64, ...), one that succeeds lifts the suspension
  int q8_suspended_searches = 0;           // eligible searches that ran on the fp16 rows since the suspension / the last failed probe
  int q8_probe_after = 0;
  bool q8_build_due = false;               // an enqueued (begin / finish) search wanted the int8 copy: _finish builds it
  // an index whose rows change between searches (add, search, add, search, ...) would pay three passes over all rows per
  // search for a copy that serves one: after two builds in a row that served at most one search each, automatic mode
  // leaves the first search after a change on the fp16 rows and rebuilds when a second search finds the rows unchanged
  int q8_searches_on_copy = 0;             // searches that scanned the current copy
  int q8_short_lived_builds = 0;           // consecutive builds whose copy served <= 1 search
  uint64_t q8_seen_epoch = 0;              // rows_epoch of the last search that found the copy stale
  int nominate_mode = 1;                   // 0 off, 1 automatic, 2 always (no profitability check); proqa_index_configure_nomination
  bool q8_active = false;                  // the search being enqueued runs its rounds on the int8 copy
  int64_t pending_nq = 0;                  // queries of the search being enqueued (the row-split launch of small batches reads it)
  double page_growth = 2.0;                // growth cap of the rounds being enqueued (the nominating merge's size follows it)
  // the 1024-key nominating merge (eight workgroups per CU) is sized for rows that nominate ~2 x their candidates; rows that
  // nominate more overflow it although the 2048-key merge would hold them: the first such search switches this index to the
  // larger merge instead of suspending the int8 rounds (until the rows change)
  bool small_merge_ok = true;
  bool used_small_merge = false;           // a round of the search being enqueued took the 1024-key merge
  unsigned overflow_bits = 0;              // OR of the overflow words of the search's rounds (bit 1 alone: only a merge's capacity)
  // Leaping rounds (plan_leap below): thresholds taken at rank j < k of the running list, a quarter of the candidates per row
  // scanned, so half the rounds or fewer; a round in which fewer than k rows reach its threshold is re-scanned by the
  // overflow-safe path (rows in an order the first rows do not stand for).  Such a search pauses the leaps of this index for
  // leap_pause searches (16, then doubling up to 1024; a search that leaps cleanly clears it)
  int leap_mode = 1;                       // 0 never, 1 automatic (developer switch PROQA_LEAP)
  int leap_pause = 0;                      // length of the current pause
  int leap_skip = 0;                       // searches of that pause still to go
  int leap_strikes = 0;                    // what the recent shortfalls cost (note_leap): a pause at eight
  bool leap_active = false;                // the search being enqueued leaps
  long long leap_key[6] = {0, 0, 0, 0, 0, 0};   // what the kept plan was made for
  int leap_plan_rounds = 0, leap_plan_rank = 0;
  double leap_plan_per_round = 0.0;
  long long leap_sample_key[4] = {0, 0, 0, 0};   // the same for the sample rounds of the one-pass search of a large k
  int leap_sample_rounds = 0, leap_sample_rank = 0;
  int leap_logged = 0;                     // the plan last reported under PROQA_LOG
  uint64_t leap_epoch = 0;                 // rows_epoch the pause belongs to (changed rows start afresh)
  int round_next_rank = 0, round_leap_check = 0, round_bit = 0;   // MergeArgs of the round being enqueued
  // queries a leaping round left short (at most kRescueMax of them, nothing else overflowed) are searched again on ordinary
  // rounds as a small batch of their own, instead of re-scanning the flagged slabs for every query (rescue_short_queries)
  unsigned* short_rounds = nullptr;        // workspace [ws_nq_pad]: bit r = fell short in round r
  struct Rescue {
    std::vector<int> ids;                  // the short queries of the search that just completed (empty: nothing to do)
    float* D = nullptr;
    long long* I = nullptr;
    int k = 0, out_stride = 0;
    long long idx_offset = 0;
    hipStream_t st = nullptr;
  } rescue;
  // compact lists the re-scoring of an int8 one-pass launch writes for its merge (one_pass_big_launch_i8)
  proqa::WaveRecord* emit_log = nullptr;
  unsigned* emit_cnt = nullptr;
  size_t emit_records = 0, emit_counts = 0;
  void* rescue_buf = nullptr;              // device: ids, fp16 query rows, D and I rows of the rescue batch
  size_t rescue_bytes = 0;
  signed char* xq8 = nullptr;              // workspace [ws_nq_pad,128]
  proqa::NominateParams* qparams = nullptr;   // workspace [ws_nq_pad]
  unsigned long long* stat_nom = nullptr;  // workspace [ws_nq_pad] rows re-scored per query
  // tuning
  int first_slab_rows = 256;
  int growth = 0;                          // 0 = automatic (see growth_for)
  proqa_search_stats stats = {};
  // a search enqueued by proqa_index_search_begin_device whose host-side completion is still due
  struct Pending {
    bool active = false;
    int qw = 0;
    unsigned n_qtiles = 0;
    int64_t nq = 0, nq_pad = 0;
    int k = 0, dtype = 0;
    const void* xq_dev = nullptr;
    float* D = nullptr;
    long long* I = nullptr;
    long long idx_offset = 0;
    hipStream_t st = nullptr;
    uint32_t* status_dev = nullptr;
    std::vector<proqa::Slab> slabs;
    long long boot = 0;
  } pending;
};

namespace proqa {
namespace {

constexpr int kMaxRounds = kOverflowWords;   // 96

// developer switches, read once: PROQA_DEBUG_CAND prints the cumulative candidate count after every
// round (adds a host sync per round), PROQA_DEBUG_ROUNDS prints per-round filter times when profiling
bool debug_flag(const char* name) { return getenv(name) != nullptr; }
const bool kDebugCand = debug_flag("PROQA_DEBUG_CAND");
// query blocks per wave for batches of more than 256 queries: 2 (8 waves x 64 queries) or, PROQA_FILTER_QW=4, the
// 4-wave variant with 128 resident queries per wave (one wave per SIMD; experiment of DESIGN.md section 2.3)
const unsigned kFilterFlags = getenv("PROQA_FILTER_FLAGS") ? (unsigned)atoi(getenv("PROQA_FILTER_FLAGS")) : 0u;
const int kWideQw = getenv("PROQA_FILTER_QW") && (atoi(getenv("PROQA_FILTER_QW")) == 4 || atoi(getenv("PROQA_FILTER_QW")) == 1)
                        ? atoi(getenv("PROQA_FILTER_QW")) : 2;
const bool kDebugRounds = debug_flag("PROQA_DEBUG_ROUNDS");
// developer A/B switches of the HBM-bound int8 scan (<= 256 queries): the one-workgroup-per-CU form with four pair buffers
// (measured slower: off), and the row-split launch of <= 128 queries (on)
const bool kDeepRing = getenv("PROQA_I8_DEEP_RING") && atoi(getenv("PROQA_I8_DEEP_RING")) != 0;
const bool kRowSplit = !(getenv("PROQA_I8_ROW_SPLIT") && atoi(getenv("PROQA_I8_ROW_SPLIT")) == 0);
// k x growth up to which a nominating round takes the 1024-key merge (developer / test override: PROQA_NOM_SMALL_MERGE_LIMIT)
const double kSmallMergeLimit = getenv("PROQA_NOM_SMALL_MERGE_LIMIT") ? atof(getenv("PROQA_NOM_SMALL_MERGE_LIMIT")) : 200.0;

// counters of the last float32 -> fp16 conversion: {values fp16 cannot hold exactly, values beyond its range}
int read_inexact(proqa_index* idx, const char* what, hipStream_t st, unsigned* n_inexact) {
  unsigned c[2] = {0, 0};
  PROQA_HIP(hipMemcpyAsync(c, idx->inexact, sizeof c, hipMemcpyDeviceToHost, st));
  PROQA_HIP(hipStreamSynchronize(st));
  if (c[1])
    return fail(PROQA_EINVAL, "%s: %u float32 values exceed the fp16 range (|x| > 65504); such embeddings are not "
                              "supported", what, c[1]);
  *n_inexact = c[0];
  return PROQA_OK;
}

int reserve_rows32(proqa_index* idx, int64_t rows) {
  if (rows <= idx->capacity32) return PROQA_OK;
  int64_t cap = std::max<int64_t>(rows, std::max(idx->capacity, idx->capacity32 + idx->capacity32 / 2));
  cap = round_up<int64_t>(cap, kStageRows);
  float* p = nullptr;
  hipError_t e = try_malloc((void**)&p, (size_t)cap * kDim * 4);
  if (e != hipSuccess) return fail(PROQA_ENOMEM, "hipMalloc of %lld float32 index rows failed: %s", (long long)cap,
                                   hipGetErrorString(e));
  // every float32 row written so far moves along -- idx->n counts the rows of COMPLETED calls only, the earlier pieces of
  // the call in progress sit above it (a reset() index re-filled by one large inexact float32 add used to lose them here)
  // (nothing is kept while the index is being switched to exact mode: enable_exact rewrites every row)
  const int64_t keep = std::min(idx->capacity32, idx->exact ? std::max(idx->n, idx->rows32) : idx->rows32);
  if (keep > 0 && idx->xb32)
    PROQA_HIP(hipMemcpy(p, idx->xb32, (size_t)keep * kDim * 4, hipMemcpyDeviceToDevice));
  if (idx->xb32) PROQA_HIP(hipFree(idx->xb32));
  idx->xb32 = p;
  idx->capacity32 = cap;
  return PROQA_OK;
}

// Switch the index to exact-float32 mode: float32 copies of the rows already stored (exact upcasts of
// their fp16 values) and the norm statistics the error margin is built from.
int enable_exact(proqa_index* idx, hipStream_t st) {
  if (idx->exact) return PROQA_OK;
  PROQA_HIP(hipStreamSynchronize(st));
  if (!idx->norm_stats) PROQA_HIP(hipMalloc((void**)&idx->norm_stats, 2 * sizeof(unsigned)));
  PROQA_HIP(hipMemsetAsync(idx->norm_stats, 0, 2 * sizeof(unsigned), st));
  idx->rows32 = 0;   // whatever an earlier exact episode left in the buffer is stale
  if (int rc = reserve_rows32(idx, std::max<int64_t>(idx->n, 1))) return rc;
  PROQA_HIP(launch_upconvert_f16_to_f32(idx->xb, idx->xb32, idx->n * kDim, st));
  PROQA_HIP(launch_row_norm_stats(idx->xb32, idx->xb, idx->n, idx->norm_stats, st));
  idx->rows32 = idx->n;
  idx->exact = true;
  return PROQA_OK;
}

// rows [r0, r0+m) were just written to xb (fp16); in exact mode give them float32 copies (from
// `src32` when the caller had float32 data, else exact upcasts) and fold them into the statistics
int finish_rows_exact(proqa_index* idx, int64_t r0, int64_t m, const float* src32_dev, hipStream_t st) {
  if (!idx->exact || m == 0) return PROQA_OK;
  if (int rc = reserve_rows32(idx, r0 + m)) return rc;
  float* dst = idx->xb32 + (size_t)r0 * kDim;
  if (src32_dev)
    PROQA_HIP(hipMemcpyAsync(dst, src32_dev, (size_t)m * kDim * 4, hipMemcpyDeviceToDevice, st));
  else
    PROQA_HIP(launch_upconvert_f16_to_f32(idx->xb + (size_t)r0 * kDim * 2, dst, m * kDim, st));
  PROQA_HIP(launch_row_norm_stats(dst, idx->xb + (size_t)r0 * kDim * 2, m, idx->norm_stats, st));
  idx->rows32 = std::max(idx->rows32, r0 + m);
  return PROQA_OK;
}

int reserve_rows(proqa_index* idx, int64_t rows) {
  if (rows <= idx->capacity) return PROQA_OK;
  if (!idx->owns_xb) return fail(PROQA_EINVAL, "index adopted caller memory; cannot grow it");
  int64_t cap = std::max<int64_t>(rows, idx->capacity + idx->capacity / 2);
  cap = round_up<int64_t>(cap, kStageRows);
  char* p = nullptr;
  hipError_t e = try_malloc((void**)&p, (size_t)cap * kDim * 2);
  if (e != hipSuccess) return fail(PROQA_ENOMEM, "hipMalloc of %lld index rows failed: %s", (long long)cap,
                                   hipGetErrorString(e));
  if (idx->n > 0) PROQA_HIP(hipMemcpy(p, idx->xb, (size_t)idx->n * kDim * 2, hipMemcpyDeviceToDevice));
  if (idx->xb) PROQA_HIP(hipFree(idx->xb));
  idx->xb = p;
  idx->capacity = cap;
  return PROQA_OK;
}

int ensure_stage(proqa_index* idx, size_t bytes) {
  if (bytes > idx->stage_bytes) {
    if (idx->stage_dev) PROQA_HIP(hipFree(idx->stage_dev));
    idx->stage_dev = nullptr;
    idx->stage_bytes = 0;
    hipError_t e = try_malloc(&idx->stage_dev, bytes);
    if (e != hipSuccess) return fail(PROQA_ENOMEM, "hipMalloc staging %zu B: %s", bytes, hipGetErrorString(e));
    idx->stage_bytes = bytes;
  }
  return PROQA_OK;
}

// The int8 copy of the rows for the nomination scan, (re)built when the rows changed since it was made: column statistics
// (two launches), the quantisation pass, one host read of the statistics word.  18M rows: ~3 passes over 4.6 GB.
constexpr int64_t kNominateMinRows = 65536;   // smaller shards are launch-bound either way
constexpr int kNominateMaxK = 128;            // nominations of a round (~3 x its candidates) must fit one merge
constexpr unsigned kNominateLaneCap = 32;     // records per lane list of an int8 round (no spill log: a full list re-scans the round)
// may the rounds of a top-k search of this index scan an int8 copy at all (mode, precision mode, k, shard size)?
bool nomination_eligible(const proqa_index* idx, int k) {
  const bool always = idx->nominate_mode == 2;
  return idx->nominate_mode != 0 && !idx->exact && k <= kNominateMaxK && idx->n >= (always ? 4 * kStageRows : kNominateMinRows);
}
int ensure_q8(proqa_index* idx, hipStream_t st) {
  if (idx->q8_epoch == idx->rows_epoch) return PROQA_OK;
  idx->q8_usable = false;
  idx->q8_unprofitable = false;
  idx->q8_suspended_searches = 0;
  idx->q8_probe_after = 0;
  idx->q8_build_due = false;
  if (idx->q8_epoch != 0) idx->q8_short_lived_builds = idx->q8_searches_on_copy <= 1 ? idx->q8_short_lived_builds + 1 : 0;
  idx->q8_searches_on_copy = 0;
  idx->small_merge_ok = true;
  if (!idx->col) {
    PROQA_HIP(hipMalloc((void**)&idx->col, 3 * kDim * sizeof(float)));
    PROQA_HIP(hipMalloc((void**)&idx->col_partial, (size_t)kColStatGroups * 3 * kDim * sizeof(float)));
    PROQA_HIP(hipMalloc((void**)&idx->qstats, sizeof(QuantStats)));
  }
  if (idx->n > idx->capacity8) {
    PROQA_HIP(hipStreamSynchronize(st));
    if (idx->xb8) PROQA_HIP(hipFree(idx->xb8));
    if (idx->blk8) PROQA_HIP(hipFree(idx->blk8));
    idx->xb8 = nullptr;
    idx->blk8 = nullptr;
    idx->capacity8 = 0;
    const int64_t cap = round_up<int64_t>(std::max(idx->n, idx->capacity), kStageRows);
    if (try_malloc((void**)&idx->xb8, (size_t)cap * kDim) != hipSuccess ||
        try_malloc((void**)&idx->blk8, (size_t)(cap / 32 + 16) * sizeof(float2)) != hipSuccess) {
      if (idx->rescue_buf) (void)hipFree(idx->rescue_buf);
  if (idx->emit_log) (void)hipFree(idx->emit_log);
  if (idx->emit_cnt) (void)hipFree(idx->emit_cnt);
  if (idx->xb8) (void)hipFree(idx->xb8);
      idx->xb8 = nullptr;
      idx->q8_epoch = idx->rows_epoch;   // no room for the copy: this index is searched on its fp16 rows
      return PROQA_OK;
    }
    idx->capacity8 = cap;
  }
  PROQA_HIP(hipMemsetAsync(idx->qstats, 0, sizeof(QuantStats), st));
  PROQA_HIP(launch_column_stats(idx->xb, idx->n, idx->col_partial, idx->col, idx->qstats, st));
  PROQA_HIP(launch_quantise_rows_i8(idx->xb, idx->n, idx->col, idx->xb8, idx->blk8, idx->qstats, st));
  QuantStats h;
  PROQA_HIP(hipMemcpyAsync(&h, idx->qstats, sizeof h, hipMemcpyDeviceToHost, st));
  PROQA_HIP(hipStreamSynchronize(st));
  idx->q8_epoch = idx->rows_epoch;
  idx->q8_usable = h.nonfinite == 0u && h.max_inorm != 0u;   // (all rows equal to the mean: nothing to scan for)
  return PROQA_OK;
}

constexpr int64_t kAddPieceRows = 1 << 20;   // rows per upload piece of proqa_index_add (256 MiB of fp16 rows)
constexpr int kLoaderSlots = 4;               // pinned pieces of proqa_index_add_npy: two being read, two on their way up
// pieces of 4 .. 64 MiB and 2 .. 8 readers all reach 31-34 GB/s from a warm page cache (profiles/r04_loader_timing.txt)
constexpr int64_t kLoaderPieceBytes = 8 << 20;

// One piece of float32 rows from host memory -> fp16 rows [row, row+m) of the index, with the exact-float32 bookkeeping.
// Rows [idx->n, row) are the earlier pieces of the same call (idx->n counts the rows of completed calls only).
int ingest_f32_piece(proqa_index* idx, int64_t row, int64_t m, const void* src_host, bool pinned, hipStream_t st) {
  if (int rc = ensure_stage(idx, (size_t)kAddPieceRows * kDim * 4)) return rc;
  char* dst = idx->xb + (size_t)row * kDim * 2;
  if (pinned)
    PROQA_HIP(hipMemcpyAsync(idx->stage_dev, src_host, (size_t)m * kDim * 4, hipMemcpyHostToDevice, st));
  else
    PROQA_HIP(hipMemcpy(idx->stage_dev, src_host, (size_t)m * kDim * 4, hipMemcpyHostToDevice));
  PROQA_HIP(hipMemsetAsync(idx->inexact, 0, 2 * sizeof(unsigned), st));
  PROQA_HIP(launch_convert_f32_to_f16((const float*)idx->stage_dev, dst, m * kDim, idx->inexact, st));
  unsigned bad = 0;
  if (int rc = read_inexact(idx, "index_add", st, &bad)) return rc;
  if (bad && !idx->allow_rounding && !idx->exact) {
    // first values fp16 cannot hold: from here on the index keeps float32 copies.  enable_exact covers the
    // rows counted in idx->n; the pieces of THIS call that were already uploaded are caught up here
    if (int rc = enable_exact(idx, st)) return rc;
    if (int rc = finish_rows_exact(idx, idx->n, row - idx->n, nullptr, st)) return rc;   // earlier pieces were exact
  }
  if (int rc = finish_rows_exact(idx, row, m, (const float*)idx->stage_dev, st)) return rc;
  if (idx->exact) PROQA_HIP(hipStreamSynchronize(st));   // stage_dev is reused by the next piece
  return PROQA_OK;
}

// state shared by the reader threads of proqa_index_add_npy and the uploading thread; the destructor stops and joins
// the readers before anything they touch goes away
struct NpyRing {
  int fd = -1;
  char* pinned = nullptr;
  hipStream_t stream = nullptr;                        // uploads that read the ring
  hipEvent_t uploaded[kLoaderSlots] = {};
  int n_slots = 0;
  int64_t n_pieces = 0;
  std::mutex m;
  std::condition_variable cv;
  int64_t next_piece = 0;                              // next piece a reader takes
  int64_t retired = 0;                                 // pieces (in order) whose upload has completed
  int64_t filled[kLoaderSlots] = {-1, -1, -1, -1};     // piece that sits in the slot, read completely
  bool failed = false;
  char why[256] = {0};
  std::vector<std::thread> threads;
  void set_error(const char* msg) {
    std::lock_guard<std::mutex> lk(m);
    if (!failed && msg) snprintf(why, sizeof why, "%s", msg);
    failed = true;
    cv.notify_all();
  }
  ~NpyRing() {
    set_error(nullptr);
    for (auto& t : threads)
      if (t.joinable()) t.join();
    if (fd >= 0) close(fd);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& e : uploaded)
      if (e) (void)hipEventDestroy(e);
    if (pinned) (void)hipHostFree(pinned);
  }
};

void free_store(proqa_index* idx) {
  void* ptrs[] = {idx->lane_log, idx->lane_cnt, idx->spill_log, idx->spill_cnt};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  idx->lane_log = nullptr;
  idx->lane_cnt = nullptr;
  idx->spill_log = nullptr;
  idx->spill_cnt = nullptr;
  idx->store_records = 0;
  idx->store_lists = 0;
  idx->store_slots = 0;
}

void free_workspace(proqa_index* idx) {
  void* ptrs[] = {idx->xq_pad, idx->tau, idx->run_n, idx->run_keys, idx->stat_dev, idx->bound_keys, idx->ub, idx->done,
                  idx->xq32, idx->margin, idx->tau_filter, idx->ub_filter, idx->xq8, idx->qparams, idx->stat_nom, idx->short_rounds};
  idx->short_rounds = nullptr;
  idx->xq8 = nullptr;
  idx->qparams = nullptr;
  idx->stat_nom = nullptr;
  idx->xq32 = nullptr;
  idx->margin = nullptr;
  idx->tau_filter = nullptr;
  idx->ub_filter = nullptr;
  idx->bound_keys = nullptr;
  idx->ub = nullptr;
  idx->done = nullptr;
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  idx->xq_pad = nullptr;
  idx->tau = nullptr;
  idx->run_n = nullptr;
  idx->run_keys = nullptr;
  idx->stat_dev = nullptr;
  idx->ws_nq_pad = 0;
  idx->ws_k = 0;
  free_store(idx);
}

int ensure_workspace(proqa_index* idx, int64_t nq_pad, int k) {
  if (nq_pad <= idx->ws_nq_pad && k <= idx->ws_k) return PROQA_OK;
  const int64_t q = std::max(nq_pad, idx->ws_nq_pad);
  const int kk = std::max(k, idx->ws_k);
  free_workspace(idx);
  PROQA_HIP(hipMalloc(&idx->xq_pad, (size_t)q * kDim * 2));
  PROQA_HIP(hipMalloc((void**)&idx->tau, (size_t)q * sizeof(float)));
  PROQA_HIP(hipMalloc((void**)&idx->run_n, (size_t)q * sizeof(unsigned)));
  PROQA_HIP(hipMalloc((void**)&idx->run_keys, (size_t)q * kk * sizeof(unsigned long long)));
  PROQA_HIP(hipMalloc((void**)&idx->stat_dev, (size_t)q * sizeof(unsigned long long)));
  PROQA_HIP(hipMalloc((void**)&idx->bound_keys, (size_t)q * sizeof(unsigned long long)));
  PROQA_HIP(hipMalloc((void**)&idx->ub, (size_t)q * sizeof(float)));
  PROQA_HIP(hipMalloc((void**)&idx->done, (size_t)q));
  PROQA_HIP(hipMalloc((void**)&idx->xq32, (size_t)q * kDim * sizeof(float)));
  PROQA_HIP(hipMalloc((void**)&idx->margin, (size_t)q * sizeof(float)));
  PROQA_HIP(hipMalloc((void**)&idx->tau_filter, (size_t)q * sizeof(float)));
  PROQA_HIP(hipMalloc((void**)&idx->ub_filter, (size_t)q * sizeof(float)));
  PROQA_HIP(hipMalloc((void**)&idx->xq8, (size_t)q * kDim));
  PROQA_HIP(hipMalloc((void**)&idx->qparams, (size_t)q * sizeof(NominateParams)));
  PROQA_HIP(hipMalloc((void**)&idx->stat_nom, (size_t)q * sizeof(unsigned long long)));
  PROQA_HIP(hipMalloc((void**)&idx->short_rounds, (size_t)q * sizeof(unsigned)));
  idx->ws_nq_pad = q;
  idx->ws_k = kk;
  return PROQA_OK;
}

// Candidate store for a launch of `chunks` corpus chunks x `n_qtiles` query tiles of `nq_pad` padded queries with
// `lane_cap` records per lane list.  The lists are addressed with the LAUNCH's own query count (CandidateStore::nq_pad),
// so the store is sized by what a launch needs, not by the largest batch the handle ever saw: it only grows, by record /
// list / slot counts.
int ensure_store(proqa_index* idx, unsigned chunks, unsigned n_qtiles, int64_t nq_pad, unsigned lane_cap) {
  const size_t lists = (size_t)chunks * nq_pad * 2;
  const size_t records = lists * lane_cap;
  const size_t slots = (size_t)chunks * n_qtiles * kFilterWaves;
  if (records <= idx->store_records && lists <= idx->store_lists && slots <= idx->store_slots) return PROQA_OK;
  const size_t want_records = std::max(records, idx->store_records), want_lists = std::max(lists, idx->store_lists),
               want_slots = std::max(slots, idx->store_slots);
  free_store(idx);
  // developer/test switch, read on this (rare) growth path only: a store above the limit fails the way a full GPU fails
  // it -- by a hipMalloc that really fails (1 EiB), so that HIP's sticky error state is the real one
  const char* lim = getenv("PROQA_DEBUG_STORE_LIMIT_MB");
  const bool refuse = lim && want_records * sizeof(WaveRecord) > ((size_t)atoll(lim) << 20);
  hipError_t e = try_malloc((void**)&idx->lane_log, refuse ? (size_t)1 << 60 : want_records * sizeof(WaveRecord));
  if (e != hipSuccess)
    return fail(PROQA_ENOMEM, "search: candidate store of %zu lists x %u records: %s (fewer queries per call need less)", lists,
                lane_cap, hipGetErrorString(e));
  idx->store_records = want_records;
  PROQA_HIP(hipMalloc((void**)&idx->lane_cnt, want_lists * sizeof(unsigned)));
  idx->store_lists = want_lists;
  PROQA_HIP(hipMalloc((void**)&idx->spill_log, want_slots * kSpillCap * sizeof(WaveRecord)));
  PROQA_HIP(hipMalloc((void**)&idx->spill_cnt, want_slots * sizeof(unsigned)));
  idx->store_slots = want_slots;
  return PROQA_OK;
}

// Geometric slab schedule over [0, n).  After `seen` rows the threshold is the k-th best of them,
// so a slab of g*seen rows yields ~g*k candidates per query; g is capped so that those fit the
// lane lists and one LDS merge pass next to the k running keys (kCandidateBudget per round).
constexpr double kCandidateBudget = 640.0;
// developer override of the budget (schedule experiments)
const double kBudget = getenv("PROQA_CAND_BUDGET") ? atof(getenv("PROQA_CAND_BUDGET")) : kCandidateBudget;

// Default growth: 4 for MFMA-bound batches (fewer candidates per round keep the rare path rare); 8 for
// the HBM-bound small batches (one query tile per wave: two rounds fewer of launch + merge latency,
// measured 6-9 % on the whole search at Q <= 256, 3 % slower at Q >= 1024).
// Rounds on the int8 copy: 2.  A nominated row costs a 256-byte gather and a share of an MFMA in the merge, and the scan
// nominates ~1.9 x the rows that pass, so the time of a search follows the nominations more steeply than an fp16 search
// follows its candidates; two or three more rounds (each ~35 us of launch + merge latency) are cheaper than the rows
// they save (scripts/dev_schedule_sweep.py, 1 .. 2032 queries x 2.25M .. 18M rows: ABLATIONS R5.8).
// Small batches (qw == 1) on the int8 copy: 6 -- their merges are few workgroups on an idle chip, what they save is the ~9 us
// every launch costs before it streams (re-swept in round 6 with the 1024-thread merges, stream time for growth 4 / 6 / 8:
// one question 0.502 / 0.488 / 0.499 ms at 18M rows, 32 queries 0.522 / 0.516 / 0.517, 128 queries 0.585 / 0.581 / 0.589; 2032
// queries stay at 2: 4.271 against 4.299 / 4.329 ms for 3 / 4: ABLATIONS R6.10).
double growth_for(int k, int configured, int qw, bool nominating = false) {
  const int g = configured > 0 ? configured : nominating ? (qw == 1 ? 6 : 2) : (qw == 1 ? 8 : 4);
  // big pages: 60 % of the free keys of the big merge (the rest is headroom for the spread of the candidate count)
  const double budget = k <= kPageK ? kBudget : 0.6 * (kBigSortKeys - k);
  return std::min<double>(g, budget / k);
}

// `start` > 0: rows [0, start) were covered by the bootstrap
// Rounds of EQUAL growth behind a bootstrap of `start` rows: as many rounds as the capped growth needs, each multiplying
// the rows seen by the same factor (N / start)^(1/R).  A schedule of "cap, cap, ..., whatever is left" wastes candidates:
// a round yields ~k x (slab / rows seen) of them whatever its size, and the time of a search follows their total
// (measured: ~17 us per 100 candidates per query at 2032 queries), so the same number of rounds at the smallest equal
// growth is the cheapest (18M rows: 6 rounds at 3.05 instead of 4, 4, 4, 4, 4, 0.4: 10 % fewer candidates).
const bool kEqualGrowth = !(getenv("PROQA_EQUAL_GROWTH") && atoi(getenv("PROQA_EQUAL_GROWTH")) == 0);   // developer A/B switch
std::vector<Slab> plan_slabs_equal(long long n, long long start, double growth_cap) {
  std::vector<Slab> out;
  if (start <= 0 || start >= n) return out;
  const double ratio = (double)n / (double)start;
  int rounds = std::max(1, (int)std::ceil(std::log(ratio) / std::log(1.0 + growth_cap) - 1e-9));
  const double step = std::pow(ratio, 1.0 / rounds);
  long long seen = start;
  for (int r = 0; r < rounds && seen < n; ++r) {
    long long r1 = r + 1 == rounds ? n : round_up<long long>((long long)(start * std::pow(step, r + 1)), kStageRows);
    r1 = std::min(n, std::max(r1, seen + kStageRows));
    out.push_back({seen, r1});
    seen = r1;
  }
  if (seen < n) out.push_back({seen, n});
  return out;
}

// developer experiment: PROQA_GROWTH_LIST="7,5,3,2" = slab / rows-seen ratio of round 1, 2, ... (the last one repeats)
const std::vector<double> kGrowthList = [] {
  std::vector<double> v;
  if (const char* e = getenv("PROQA_GROWTH_LIST")) {
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const double g = strtod(p, &end);
      if (end == p) break;
      if (g > 0) v.push_back(g);
      p = *end ? end + 1 : end;
    }
  }
  return v;
}();

std::vector<Slab> plan_slabs(long long n, int first, double growth, long long start = 0) {
  std::vector<Slab> out;
  long long seen = start;
  auto growth_of = [&](size_t round) {
    if (kGrowthList.empty() || start == 0) return growth;
    return std::min(growth * 2.0, kGrowthList[std::min(round, kGrowthList.size() - 1)]);
  };
  long long next = start > 0 ? std::max<long long>(kStageRows, round_up<long long>((long long)(start * growth_of(0)), kStageRows))
                             : std::min<long long>(n, round_up<long long>(first, kStageRows));
  while (seen < n) {
    long long r1 = std::min(n, seen + next);
    // do not leave a tiny tail for an extra round
    if (n - r1 < next / 4) r1 = n;
    out.push_back({seen, r1});
    seen = r1;
    next = std::max<long long>(kStageRows, round_up<long long>((long long)(seen * growth_of(out.size())), kStageRows));
  }
  return out;
}

// ---- leaping rounds ------------------------------------------------------------------------------------------------
// A round that tests against the k-th best score of the n0 rows seen so far logs ~k (rho - 1) rows of the next (rho - 1) n0,
// which is what caps the growth of the ordinary schedule (a merge holds so many keys; every candidate is a gather).  The
// rows that end up in the best k are far fewer: a round may test against the score at rank j < k instead -- it then logs
// ~j (rho - 1) rows -- provided at least k rows of the rho n0 now seen BEAT that score, for then every row of the best k
// does, and each of them is in the running list (an earlier row: the list holds the best k of those) or was logged.  The
// merge verifies exactly that (its k-th key beats the threshold it was handed, topk_merge: note_rank) and marks the queries
// for which it does not hold; those are searched again on ordinary rounds (page_complete -> rescue_short_queries), or, when a
// list or a merge overflowed as well, the flagged slabs are re-scanned against the k-th best scores like any overflowed
// round.  For rows in an order the first n0 stand for (exchangeable: no sorting by topic or norm, no long runs of rows that
// score alike), the count of new rows above the rank-j score is negative-binomial (j, 1 / rho): mean j (rho - 1), and
// P(fewer than k - j) is a closed sum -- leap_rank takes the smallest j that keeps it below kLeapEps per query and round
// (1e-8: one second search in ~10^4 searches of 2032 queries).  The result never depends on any of this.
// (developer switches, read at every search so that one process can alternate them: PROQA_LEAP=0 never; PROQA_LEAP_ROUNDS /
// PROQA_LEAP_RANK / PROQA_LEAP_EPS fix the rounds behind the bootstrap, the rank, the probability)
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
constexpr int kLeapMaxK = 128;
constexpr double kLeapEps = 1e-8;
// Queries that leaping rounds left short are searched again by themselves -- all of them, however many: a second search of s <= nq
// queries on ordinary rounds never costs more than the first one did, the overflow-safe re-scan of the flagged slabs for
// every query ~1.7 x (developer / test switch PROQA_LEAP_RESCUE_MAX: at most that many, 0 = always the slab re-scan).  Up to
// kRescueCheap of them count as a cheap shortfall (note_leap).
constexpr int kRescueCheap = 256;
const int kRescueMax = getenv("PROQA_LEAP_RESCUE_MAX") ? atoi(getenv("PROQA_LEAP_RESCUE_MAX")) : (1 << 30);

// P(fewer than k rows beat the rank-j score after the rows seen grew by the factor rho): the negative-binomial sum
double leap_fail_probability(int k, int j, double rho) {
  const double p = 1.0 / rho;
  double pmf = std::pow(p, j), sum = 0.0;   // P(X = 0)
  for (int x = 0; x < k - j; ++x) {
    sum += pmf;
    pmf *= (double)(x + j) / (double)(x + 1) * (1.0 - p);
  }
  return sum;
}
int leap_rank(int k, double rho, double eps) {
  for (int j = 1; j < k; ++j)
    if (leap_fail_probability(k, j, rho) <= eps) return j;
  return k;
}
struct LeapPlan {
  int rounds = 0;   // 0: no leap
  int rank = 0;
  double per_round = 0.0;   // rows expected to beat the threshold per round and query: rank (rho - 1)
};
// The schedule: `rounds` rounds of equal growth behind the bootstrap, the one that minimises rounds x (c + rows per round),
// c = what a round costs beside the rows it logs, in rows per query: ~50 for a batch that fills the chip, ~120 for the
// batches of <= 256 queries, whose rounds are launch latencies (interleaved sweeps, scripts/dev_leap_sweep.py, ABLATIONS R6.13:
// 2032 queries: 4-5 rounds at 18M rows, 3 at 2.25M; <= 256 queries: 3 and 2).  The rows per round are capped by what the
// round's merge holds: their count varies by a relative 1 / sqrt(rank) around rank (rho - 1) (5 sigma are allowed for), and the
// int8 scan nominates ~2-2.5 x the rows that pass (2048-key merge: 2048 nominated rows).
LeapPlan plan_leap(long long n, long long boot, int k, int qw, bool nominating, int max_k = kLeapMaxK) {
  LeapPlan best;
  if (boot <= 0 || n <= boot || k > max_k || k < 8) return best;
  const int fixed_rounds = env_int("PROQA_LEAP_ROUNDS", 0), fixed_rank = env_int("PROQA_LEAP_RANK", 0);
  const double eps = getenv("PROQA_LEAP_EPS") ? atof(getenv("PROQA_LEAP_EPS")) : kLeapEps;
  const double ratio = (double)n / (double)boot;
  const double round_cost = getenv("PROQA_LEAP_ROUND_COST") ? atof(getenv("PROQA_LEAP_ROUND_COST")) : (qw == 1 ? 120.0 : 50.0);
  double best_cost = 0.0;
  for (int r = 1; r <= 16; ++r) {
    if (fixed_rounds > 0 && r != fixed_rounds) continue;
    const double rho = std::pow(ratio, 1.0 / r);
    if (rho < 1.5) break;
    const int j = fixed_rank > 0 ? std::min(fixed_rank, k) : leap_rank(k, rho, eps);
    if (j >= k) continue;
    const double m = j * (rho - 1.0);
    const double spread = 1.0 + 5.0 / std::sqrt((double)j);
    const bool fits = nominating ? m * spread * 2.5 <= (double)kMaxSortKeys : m * spread <= (double)(kMaxSortKeys - k) && m <= kBudget;
    if (!fits && fixed_rounds <= 0) continue;
    const double cost = r * (round_cost + m);
    if (!best.rounds || cost < best_cost) {
      best.rounds = r;
      best.rank = j;
      best.per_round = m;
      best_cost = cost;
    }
  }
  return best;
}

struct LaunchGeom {
  int rows_per_chunk;
  unsigned chunks;   // chunks that scan rows
  unsigned grid;     // workgroups launched (chunks padded to a multiple of 8, times n_qtiles)
};

LaunchGeom geometry(long long slab_rows, unsigned n_qtiles, bool single_stage, int page_k, unsigned want_chunks = 0) {
  const int cus = device_cu_count();
  // big pages put thousands of candidates per query into a round: more chunks = more lane lists to spread them over
  // (the one-pass launch of a large k asks for its own count)
  const long long min_chunks = want_chunks ? want_chunks : (page_k > kPageK ? kBigMinChunks : 64);
  // one workgroup per CU: target_chunks * n_qtiles ~= #CUs, chunks a multiple of 8 (XCD map); never
  // fewer than 64 chunks, so that a round's ~kCandidateBudget records per query spread over >= 128
  // lane lists (capacity kLaneCap each) however many query tiles there are
  long long target = std::max<long long>(min_chunks, (cus / (long long)n_qtiles) / 8 * 8);
  long long rpc = round_up<long long>(ceil_div<long long>(slab_rows, target), kStageRows);
  // dense launches (threshold -inf, or the inclusive overflow-safe re-scan) log EVERY tile: a lane
  // list holds exactly one stage of them (kLaneCap = 8 tiles), so each chunk is one stage
  if (single_stage) rpc = kStageRows;
  const long long chunks = ceil_div<long long>(slab_rows, rpc);
  return {(int)rpc, (unsigned)chunks, (unsigned)(round_up<long long>(chunks, 8) * n_qtiles)};
}

CandidateStore store_of(const proqa_index* idx, unsigned nq_pad, unsigned n_qtiles, unsigned lane_cap, unsigned n_chunks) {
  CandidateStore st;
  st.lane_log = idx->lane_log;
  st.lane_cnt = idx->lane_cnt;
  st.spill_log = idx->spill_log;
  st.spill_cnt = idx->spill_cnt;
  st.nq_pad = nq_pad;
  st.n_qtiles = n_qtiles;
  st.lane_cap = lane_cap;
  st.n_chunks = n_chunks;
  return st;
}

// the one-pass launch of a large k overrides what a round derives from k
struct RoundShape {
  unsigned want_chunks = 0;   // corpus chunks to aim at
  unsigned lane_cap = 0;      // records per lane list
  int sort_cap = 0;           // keys the merge holds
  bool compact = false;       // lane lists of 8-byte keys instead of column records (the dense one-pass launch)
};

#ifdef PROQA_MERGE_STAMPS
// developer build: every 16th merge launch runs with s_memtime stamps of its phases and prints their medians
bool merge_stamps_dump(MergeArgs& ma, unsigned nq_pad, hipStream_t st) {
  static unsigned long long* dbg_buf = nullptr;
  if (!dbg_buf) (void)hipMalloc((void**)&dbg_buf, 16384 * 8 * 8);
  ma.dbg = dbg_buf;
  static int launches = 0;
  if (!(getenv("PROQA_MERGE_STAMPS_DUMP") && ++launches % 16 == 15)) return false;
  (void)launch_merge(ma, nq_pad, st);
  (void)hipStreamSynchronize(st);
  std::vector<unsigned long long> h((size_t)nq_pad * 8);
  (void)hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost);
  for (int ph = 1; ph <= 6; ++ph) {
    std::vector<long long> d;
    for (unsigned q = 0; q < nq_pad && q < 2000; ++q) d.push_back((long long)(h[q * 8 + ph] - h[q * 8 + ph - 1]));
    std::sort(d.begin(), d.end());
    fprintf(stderr, "merge phase %d: median %lld  p10 %lld  p90 %lld (s_memtime ticks)\n", ph, d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
  }
  if (ma.xb16) {   // int8 nomination rounds: the re-scoring sits inside phase 4 (stamp 7 = its end)
    std::vector<long long> d;
    for (unsigned q = 0; q < nq_pad && q < 2000; ++q) d.push_back((long long)(h[q * 8 + 7] - h[q * 8 + 3]));
    std::sort(d.begin(), d.end());
    fprintf(stderr, "  of phase 4, re-scoring of the nominated rows: median %lld  p10 %lld  p90 %lld\n", d[d.size() / 2], d[d.size() / 10],
            d[d.size() * 9 / 10]);
  }
  std::vector<long long> e;
  unsigned long long t0 = ~0ull;
  for (unsigned q = 0; q < nq_pad && q < 2000; ++q) t0 = std::min(t0, h[q * 8]);
  for (unsigned q = 0; q < nq_pad && q < 2000; ++q) e.push_back((long long)(h[q * 8] - t0));
  std::sort(e.begin(), e.end());
  fprintf(stderr, "merge start skew: median %lld max %lld\n", e[e.size() / 2], e.back());
  return true;
}
#endif

int run_round(proqa_index* idx, const Slab& slab, int qw, unsigned n_qtiles, unsigned nq_pad, int k,
              bool inclusive, bool dense, bool bounded, unsigned* overflow_word, hipStream_t st, hipEvent_t f0,
              hipEvent_t f1, const RoundShape& shape = RoundShape()) {
  // rounds of the plain search (strict threshold, first page, not dense, no shape of its own) scan the int8 copy when the
  // search was set up for it: two workgroups per CU, deep lane lists instead of a spill log
  const bool nominate = idx->q8_active && !inclusive && !dense && !bounded && !shape.want_chunks && !shape.compact;
  if (nominate) {
    // (developer switch, off: ONE workgroup per CU with four pair buffers -- 96 KB of LDS-DMA in flight -- for the batches of
    // <= 256 queries measured 10-19 % slower than two workgroups with two, ABLATIONS R6.2)
    const bool deep = qw == 1 && kDeepRing;
    // at most 128 queries: row-split launch (mips_filter_i8<SPLIT>) -- the 1 / 2 / 4 query blocks that hold queries are
    // replicated over the eight waves, which share the units of the stream (and the lists of their queries)
    unsigned q_blocks = 0;
    if (qw == 1 && kRowSplit && idx->pending_nq <= 128) q_blocks = idx->pending_nq <= 32 ? 1u : (idx->pending_nq <= 64 ? 2u : 4u);
    const unsigned want = std::max<unsigned>(64u, (unsigned)((deep ? 1 : 2) * device_cu_count() / (int)n_qtiles) / 8 * 8);
    const LaunchGeom g = geometry(slab.r1 - slab.r0, n_qtiles, false, k, want);
    if (int rc = ensure_store(idx, round_up<unsigned>(g.chunks, 8), n_qtiles, nq_pad, kNominateLaneCap)) return rc;
    FilterArgsI8 fa;
    fa.xq8 = idx->xq8;
    fa.xb8 = idx->xb8;
    fa.slab_row0 = slab.r0;
    fa.slab_row1 = slab.r1;
    fa.rows_per_chunk = g.rows_per_chunk;
    fa.tau = idx->tau;
    fa.qp = idx->qparams;
    fa.blk = idx->blk8;
    fa.store = store_of(idx, nq_pad, n_qtiles, kNominateLaneCap, round_up<unsigned>(g.chunks, 8));
    fa.overflow = overflow_word;
    fa.flags = kFilterFlags;
    fa.q_blocks = q_blocks;
    if (f0) PROQA_HIP(hipEventRecord(f0, st));
    PROQA_HIP(launch_filter_i8(fa, qw, g.grid, st, deep));
    if (f1) PROQA_HIP(hipEventRecord(f1, st));
    MergeArgs ma = {};
    ma.store = fa.store;
    ma.n_chunks = g.chunks;
    ma.qw = (unsigned)qw;
    ma.run_keys = idx->run_keys;
    ma.run_n = idx->run_n;
    ma.tau = idx->tau;
    ma.k = k;
    ma.sort_cap = sort_capacity(k);
    ma.stat_candidates = idx->stat_dev;
    ma.overflow = overflow_word;
    ma.xq16 = idx->xq_pad;
    ma.xb16 = idx->xb;
    ma.stat_nominated = idx->stat_nom;
    // a round nominates ~2 k x growth rows per query (N(0,1) data; a query's count varies by +-30 % around that): the
    // 1024-key merge (eight workgroups per CU) where that leaves a factor ~4 of headroom, else the 2048-key one
    ma.nom_keys = idx->small_merge_ok && (double)k * idx->page_growth <= kSmallMergeLimit ? 1024 : 2048;
    if (ma.nom_keys == 1024) idx->used_small_merge = true;
    ma.next_rank = idx->round_next_rank;
    ma.leap_check = idx->round_leap_check;
    ma.short_rounds = idx->short_rounds;
    ma.round_bit = idx->round_bit;
#ifdef PROQA_MERGE_STAMPS
    if (merge_stamps_dump(ma, nq_pad, st)) return PROQA_OK;
#endif
    PROQA_HIP(launch_merge(ma, nq_pad, st));
    if (kDebugCand) {   // developer: records the scan logged / rows the merges re-scored so far / candidates so far
      (void)hipStreamSynchronize(st);
      const size_t n_cnt = (size_t)fa.store.nq_pad * fa.store.n_chunks * 2;
      std::vector<unsigned> cnt(n_cnt);
      (void)hipMemcpy(cnt.data(), idx->lane_cnt, n_cnt * sizeof(unsigned), hipMemcpyDeviceToHost);
      unsigned long long recs = 0, fullest = 0;
      for (unsigned v : cnt) {
        recs += v;
        fullest = std::max<unsigned long long>(fullest, v);
      }
      std::vector<unsigned long long> nomv(nq_pad), candv(nq_pad);
      (void)hipMemcpy(nomv.data(), idx->stat_nom, nq_pad * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipMemcpy(candv.data(), idx->stat_dev, nq_pad * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long nom = 0, cand = 0;
      for (unsigned q = 0; q < nq_pad; ++q) {
        nom += nomv[q];
        cand += candv[q];
      }
      fprintf(stderr, "int8 round rows [%lld, %lld): %u chunks x %d rows, %llu records logged (fullest list %llu of %u), cumulative nominated %llu, "
              "cumulative candidates %llu; nominated q0..7:", slab.r0, slab.r1, g.chunks, g.rows_per_chunk, recs, fullest, kNominateLaneCap, nom, cand);
      for (unsigned q = 0; q < 8 && q < nq_pad; ++q) fprintf(stderr, " %llu", nomv[q]);
      fprintf(stderr, "\n");
    }
    return PROQA_OK;
  }
  const LaunchGeom g = geometry(slab.r1 - slab.r0, n_qtiles, dense, k, shape.want_chunks);
  const unsigned lane_cap = shape.lane_cap ? shape.lane_cap : lane_capacity(k);
  if (int rc = ensure_store(idx, round_up<unsigned>(g.chunks, 8), n_qtiles, nq_pad, lane_cap)) return rc;
  FilterArgs fa;
  fa.xq = idx->xq_pad;
  fa.xb = idx->xb;
  fa.slab_row0 = slab.r0;
  fa.slab_row1 = slab.r1;
  fa.rows_per_chunk = g.rows_per_chunk;
  // exact-float32 mode: the fp16 filter tests against thresholds moved by the error margin
  fa.tau = idx->exact ? idx->tau_filter : idx->tau;
  fa.ub = bounded ? (idx->exact ? idx->ub_filter : idx->ub) : nullptr;
  fa.store = store_of(idx, nq_pad, n_qtiles, lane_cap, round_up<unsigned>(g.chunks, 8));
  fa.overflow = overflow_word;
  fa.flags = kFilterFlags;
  fa.compact = shape.compact ? 1 : 0;
  if (f0) PROQA_HIP(hipEventRecord(f0, st));
  PROQA_HIP(launch_filter(fa, qw, inclusive, g.grid, st));
  if (f1) PROQA_HIP(hipEventRecord(f1, st));

  MergeArgs ma;
  ma.store = fa.store;
  ma.n_chunks = g.chunks;
  ma.qw = (unsigned)qw;
  ma.run_keys = idx->run_keys;
  ma.run_n = idx->run_n;
  ma.tau = idx->tau;
  ma.k = k;
  ma.sort_cap = shape.sort_cap ? shape.sort_cap : sort_capacity(k);
  ma.inclusive = inclusive ? 1 : 0;
  ma.bound_keys = bounded ? idx->bound_keys : nullptr;
  ma.stat_candidates = idx->stat_dev;
  ma.overflow = overflow_word;
  ma.xq32 = idx->exact ? idx->xq32 : nullptr;
  ma.xb32 = idx->exact ? idx->xb32 : nullptr;
  ma.margin = idx->exact ? idx->margin : nullptr;
  ma.tau_filter = idx->exact ? idx->tau_filter : nullptr;
  ma.dbg = nullptr;
  ma.compact = shape.compact ? 1 : 0;
  ma.xq16 = nullptr;
  ma.xb16 = nullptr;
  ma.stat_nominated = nullptr;
  ma.nom_keys = 0;
  ma.next_rank = inclusive ? 0 : idx->round_next_rank;
  ma.leap_check = inclusive ? 0 : idx->round_leap_check;
  ma.short_rounds = idx->short_rounds;
  ma.round_bit = idx->round_bit;
#ifdef PROQA_MERGE_STAMPS
  if (merge_stamps_dump(ma, nq_pad, st)) return PROQA_OK;
#endif
  PROQA_HIP(launch_merge(ma, nq_pad, st));
  return PROQA_OK;
}

// Exact top-k of rows [0, rows) for every query in two launches (see bootstrap_scores / bootstrap_select): the state the
// rounds would have after those rows.  Its overflow word is the last-but-one (the last belongs to the overflow-safe
// re-scans).
int run_bootstrap(proqa_index* idx, long long rows, unsigned nq_pad, int k, hipStream_t st, int run_stride = 0, int tau_rank = 0) {
  const size_t need = (size_t)idx->ws_nq_pad * round_up<long long>(rows, 32);
  if (need > idx->boot_floats) {
    PROQA_HIP(hipStreamSynchronize(st));
    if (idx->boot_scores) PROQA_HIP(hipFree(idx->boot_scores));
    idx->boot_scores = nullptr;
    idx->boot_floats = 0;
    PROQA_HIP(hipMalloc((void**)&idx->boot_scores, need * sizeof(float)));
    idx->boot_floats = need;
  }
  PROQA_HIP(launch_bootstrap(idx->xb, idx->xq_pad, (int)rows, nq_pad, k, idx->boot_scores, idx->run_keys, idx->run_n, idx->tau,
                             idx->stat_dev, idx->overflow + kMaxRounds - 2, st, run_stride, tau_rank));
  return PROQA_OK;
}

// One page of results (page_k <= kPageK best rows below the page bound): rounds of filter + merge
// over geometrically growing slabs, then the overflow-safe re-scan of any round that overflowed.
struct PageOut {
  float* D;
  long long* I;
  long long idx_offset;
  int out_stride;  // k of the whole search
  int out_offset;  // first result slot of this page
};

// page_enqueue puts the whole page on the stream -- bootstrap, rounds, the (optimistic) result, the copies of the overflow
// words -- and does not wait; page_complete runs after the stream has been synchronised: it re-scans what overflowed.
// `status_dev` (optional device word): 1 if page_complete is going to rewrite the result, else 0.
struct PagePlan {
  std::vector<Slab> slabs;
  long long boot = 0;
  int leap_rank = 0;   // > 0: the rounds test against that rank of the running list (plan_leap)
};

int page_enqueue(proqa_index* idx, int qw, unsigned n_qtiles, int64_t nq, int64_t nq_pad, int page_k, bool bounded,
                 const PageOut& out, hipStream_t st, bool use_bootstrap, PagePlan* plan, uint32_t* status_dev) {
  // (the round words idx->overflow were zeroed by the prep_queries launch that precedes every page)
  // Bootstrap: exact top-k of the first rows in two launches instead of the first three or four (dense) rounds.
  // Not for pages after the first (bounded), exact-float32 mode (its scores are re-computed from float32 rows),
  // k beyond the select kernel's bound, or an index too small to need it.
  // (rounds on the int8 copy nominate ~2 x the candidates of an fp16 round: growth_for gives them their own, smaller growth)
  const int gqw = qw;
  long long boot = 0;
  if (use_bootstrap && idx->bootstrap_rows > 0 && !bounded && !idx->exact && page_k <= kBootstrapMaxK &&
      page_k <= idx->bootstrap_rows / 4 && idx->n >= 4ll * idx->bootstrap_rows) {
    boot = std::min<long long>(idx->bootstrap_rows, kBootstrapMaxRows);
    // Twice the rows in the bootstrap when that does not change the number of rounds behind it: the same rounds then
    // grow by a smaller factor each, i.e. log fewer candidates (a 2.25M-row shard: 4 rounds either way, 1076 instead of
    // 1324 candidates per query, -40 us; where it would save a round instead -- 4.5M, 18M rows -- the larger bootstrap
    // costs what the round did)
    // (nominating rounds: always -- the rows of the larger bootstrap are rows no round nominates from)
    if (idx->bootstrap_auto && kEqualGrowth && boot * 2 <= kBootstrapMaxRows && idx->n >= 8 * boot) {
      const double cap = std::log(1.0 + growth_for(page_k, idx->growth, gqw, idx->q8_active));
      const int r1 = (int)std::ceil(std::log((double)idx->n / (double)boot) / cap - 1e-9);
      const int r2 = (int)std::ceil(std::log((double)idx->n / (double)(2 * boot)) / cap - 1e-9);
      if (r1 == r2 || idx->q8_active) boot *= 2;
    }
  }
  // every row of the first slab is a candidate (threshold -inf): it must fit one merge pass
  // (big pages: as many rows as the merge holds -- their growth per round is small, so the rounds should start high)
  const int first_cap = (sort_capacity(page_k) - page_k) / kStageRows * kStageRows;
  const int first = page_k > kPageK ? first_cap : std::min<int>(idx->first_slab_rows, first_cap);
  double page_growth = growth_for(page_k, idx->growth, gqw, idx->q8_active);
  // leaping rounds (plan_leap): behind a bootstrap, on the default schedule, unless this index is pausing them
  LeapPlan leap;
  if (boot && !bounded && !idx->exact && idx->leap_mode && env_int("PROQA_LEAP", 1) != 0 && idx->growth == 0 && kEqualGrowth && kGrowthList.empty() &&
      page_k <= kLeapMaxK) {
    if (idx->leap_epoch != idx->rows_epoch) {
      idx->leap_epoch = idx->rows_epoch;
      idx->leap_pause = idx->leap_skip = idx->leap_strikes = 0;
    }
    if (idx->leap_skip > 0)
      --idx->leap_skip;
    else {
      // (the plan of the last search is kept: the same index, batch class and k ask again and again)
      const long long key[6] = {idx->n, boot, page_k, qw, idx->q8_active ? 1 : 0,
                                env_int("PROQA_LEAP_ROUNDS", 0) * 1000 + env_int("PROQA_LEAP_RANK", 0)};
      if (std::memcmp(key, idx->leap_key, sizeof key) != 0 || getenv("PROQA_LEAP_EPS") || getenv("PROQA_LEAP_ROUND_COST")) {
        std::memcpy(idx->leap_key, key, sizeof key);
        const LeapPlan lp = plan_leap(idx->n, boot, page_k, qw, idx->q8_active);
        idx->leap_plan_rounds = lp.rounds;
        idx->leap_plan_rank = lp.rank;
        idx->leap_plan_per_round = lp.per_round;
      }
      leap.rounds = idx->leap_plan_rounds;
      leap.rank = idx->leap_plan_rank;
      leap.per_round = idx->leap_plan_per_round;
    }
  }
  if (leap.rounds) {
    // (equal growth over exactly leap.rounds rounds; the nominating merge is sized by the rows expected per round)
    const double rho = std::pow((double)idx->n / (double)boot, 1.0 / leap.rounds);
    plan->slabs = plan_slabs_equal(idx->n, boot, rho - 1.0 + 1e-6);
    page_growth = leap.per_round / page_k;
    plan->leap_rank = leap.rank;
    idx->leap_active = true;
    idx->stats.leap_rank = leap.rank;
    if (log_enabled() && idx->leap_logged != leap.rounds * 1000 + leap.rank) {
      idx->leap_logged = leap.rounds * 1000 + leap.rank;
      log_line("index %p: %d leaping rounds behind a bootstrap of %lld rows, thresholds at rank %d of %d (~%.0f rows per round and query "
               "reach them)", (void*)idx, leap.rounds, boot, leap.rank, page_k, leap.per_round);
    }
  } else {
    plan->slabs = boot && kEqualGrowth && kGrowthList.empty() ? plan_slabs_equal(idx->n, boot, page_growth)
                                                            : plan_slabs(idx->n, first, page_growth, boot);
  }
  idx->page_growth = page_growth;
  plan->boot = boot;
  const std::vector<Slab>& slabs = plan->slabs;
  if ((int)slabs.size() + 2 > kMaxRounds) return fail(PROQA_EINVAL, "search: too many rounds (%zu)", slabs.size());
  if (boot)
    if (int rc = run_bootstrap(idx, boot, (unsigned)nq_pad, page_k, st, 0, plan->leap_rank)) return rc;
  const bool prof = idx->profile && !bounded;  // the per-round brackets describe the first page
  for (size_t r = 0; r < slabs.size(); ++r) {
    hipEvent_t f0 = prof ? idx->ev_filter[2 * r] : nullptr;
    hipEvent_t f1 = prof ? idx->ev_filter[2 * r + 1] : nullptr;
    // while fewer than page_k rows have been merged the threshold is still -inf: every row is logged
    const bool dense = slabs[r].r0 < page_k;
    // (a leaping round's merge verifies its threshold and leaves the next round's: the same rank, the k-th best after the last)
    idx->round_leap_check = plan->leap_rank ? 1 : 0;
    idx->round_next_rank = plan->leap_rank && r + 1 < slabs.size() ? plan->leap_rank : 0;
    idx->round_bit = (int)r;
    const int rc_round = run_round(idx, slabs[r], qw, n_qtiles, (unsigned)nq_pad, page_k, false, dense, bounded,
                                   idx->overflow + r, st, f0, f1);
    idx->round_leap_check = idx->round_next_rank = 0;
    if (int rc = rc_round)
      return rc;
    if (kDebugCand) {
      (void)hipStreamSynchronize(st);
      std::vector<unsigned long long> per_query((size_t)nq);
      (void)hipMemcpy(per_query.data(), idx->stat_dev, (size_t)nq * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long c = 0;
      for (int64_t i = 0; i < nq; ++i) c += per_query[i];
      fprintf(stderr, "after round %zu: cumulative candidates %llu\n", r, c);
    }
  }
  // results are written optimistically before the one host sync of the page; they are rewritten
  // by page_complete only if a round overflowed and had to be re-scanned
  // (the finalize kernel also writes the overflow words and the candidate count into the pinned mirror and the status
  // word: no copy command between the search and what the caller enqueues behind it)
  PROQA_HIP(launch_finalize(idx->run_keys, idx->run_n, nq, page_k, out.idx_offset, out.D, out.I, out.out_stride,
                            out.out_offset, idx->overflow, status_dev, idx->mirror, idx->stat_dev, st,
                            idx->q8_active ? idx->stat_nom : nullptr));
  PROQA_HIP(hipEventRecord(idx->ev[1], st));
  return PROQA_OK;
}

// the stream has been synchronised since page_enqueue
int page_complete(proqa_index* idx, int qw, unsigned n_qtiles, int64_t nq, int64_t nq_pad, int page_k, bool bounded,
                  const PageOut& out, hipStream_t st, const PagePlan& plan, int* fallback_out, bool* bootstrap_overflow) {
  const std::vector<Slab>& slabs = plan.slabs;
  const bool prof = idx->profile && !bounded;
  if (plan.boot && idx->mirror->overflow[kMaxRounds - 2]) {   // adversarial order: the caller repeats the page without it
    *bootstrap_overflow = true;
    return PROQA_OK;
  }
  const int fallback_before = *fallback_out;
  // Overflow-safe re-scan of a round that dropped candidates (its running lists are still valid subsets and its
  // thresholds valid lower bounds).  The slab is scanned again with the inclusive threshold (rows already merged
  // come back and are de-duplicated by the merge) in four pieces, one after the other, so that each piece sees the
  // thresholds the previous ones tightened; a piece that overflows again is split further, down to sub-slabs that
  // hold fewer rows than one merge pass holds keys and are scanned densely (single-stage chunks): those cannot
  // overflow.  An unlucky batch therefore costs about one more pass over the slab, not thousands of launches;
  // only an adversarial row order reaches the dense leaves.
  const long long leaf_rows = (long long)((sort_capacity(page_k) - page_k) / kStageRows) * kStageRows;
  unsigned* word = idx->overflow + kMaxRounds - 1;
  // Leaping rounds that fell short and nothing else (no list, no merge overflowed): which queries?  They are searched
  // again as a batch of their own once this search is complete (rescue_short_queries: a few -- the usual case on rows in a
  // loose order -- are an HBM-bound small-batch search; all of them, on rows sorted against the queries, one more search on
  // ordinary rounds) instead of an fp16 pass over the flagged slabs for every query.
  bool rescued = false;
  if (idx->leap_active && !bounded && kRescueMax > 0) {
    unsigned bits = 0;
    for (size_t r = 0; r < slabs.size(); ++r) bits |= idx->mirror->overflow[r];
    if (bits == 8u) {
      std::vector<unsigned> flags((size_t)nq);
      PROQA_HIP(hipMemcpyAsync(flags.data(), idx->short_rounds, (size_t)nq * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      PROQA_HIP(hipStreamSynchronize(st));
      std::vector<int> ids;
      for (int64_t q = 0; q < nq && ids.size() <= (size_t)kRescueMax; ++q)
        if (flags[(size_t)q]) ids.push_back((int)q);
      if (!ids.empty() && ids.size() <= (size_t)kRescueMax) {
        idx->overflow_bits |= 8u;
        idx->rescue.ids.swap(ids);
        idx->rescue.D = out.D + out.out_offset;
        idx->rescue.I = out.I + out.out_offset;
        idx->rescue.k = page_k;
        idx->rescue.out_stride = out.out_stride;
        idx->rescue.idx_offset = out.idx_offset;
        idx->rescue.st = st;
        rescued = true;
      }
    }
  }
  for (size_t r = 0; r < slabs.size() && !rescued; ++r) {
    if (!idx->mirror->overflow[r]) continue;
    idx->overflow_bits |= idx->mirror->overflow[r];
    std::vector<Slab> todo;
    auto push_quarters = [&](const Slab& sl) {   // pushed in reverse: the stack pops them in row order
      const long long q = round_up<long long>(ceil_div<long long>(sl.r1 - sl.r0, 4), kStageRows);
      for (int i = 3; i >= 0; --i) {
        const long long a0 = sl.r0 + i * q, a1 = std::min(sl.r1, a0 + q);
        if (a0 < a1) todo.push_back({a0, a1});
      }
    };
    push_quarters(slabs[r]);
    while (!todo.empty()) {
      const Slab sub = todo.back();
      todo.pop_back();
      const bool leaf = sub.r1 - sub.r0 <= leaf_rows;
      if (!leaf) PROQA_HIP(hipMemsetAsync(word, 0, sizeof(unsigned), st));
      if (int rc = run_round(idx, sub, qw, n_qtiles, (unsigned)nq_pad, page_k, true, leaf, bounded, word, st, nullptr, nullptr))
        return rc;
      ++*fallback_out;
      if (leaf) continue;
      PROQA_HIP(hipMemcpyAsync(&idx->mirror->overflow[kMaxRounds - 1], word, sizeof(unsigned), hipMemcpyDeviceToHost, st));
      PROQA_HIP(hipStreamSynchronize(st));
      if (idx->mirror->overflow[kMaxRounds - 1]) push_quarters(sub);
    }
  }
  if (*fallback_out != fallback_before) {
    PROQA_HIP(launch_finalize(idx->run_keys, idx->run_n, nq, page_k, out.idx_offset, out.D, out.I, out.out_stride,
                              out.out_offset, idx->overflow, nullptr, idx->mirror, idx->stat_dev, st,
                              idx->q8_active ? idx->stat_nom : nullptr));
    PROQA_HIP(hipEventRecord(idx->ev[1], st));
    PROQA_HIP(hipStreamSynchronize(st));
  }
  idx->stats.rounds += (int)slabs.size();
  if (prof) {
    float sum = 0.f;
    for (size_t r = 0; r < slabs.size(); ++r) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, idx->ev_filter[2 * r], idx->ev_filter[2 * r + 1]);
      sum += ms;
    }
    idx->stats.filter_ms = sum;
    if (kDebugRounds) {
      for (size_t r = 0; r < slabs.size(); ++r) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, idx->ev_filter[2 * r], idx->ev_filter[2 * r + 1]);
        const LaunchGeom g = geometry(slabs[r].r1 - slabs[r].r0, n_qtiles, slabs[r].r0 < page_k, page_k);
        fprintf(stderr, "round %zu rows [%lld,%lld) grid %u rpc %d filter %.3f ms\n", r, slabs[r].r0, slabs[r].r1,
                g.grid, g.rows_per_chunk, ms);
      }
    }
  }
  return PROQA_OK;
}

int search_page(proqa_index* idx, int qw, unsigned n_qtiles, int64_t nq, int64_t nq_pad, int page_k, bool bounded,
                const PageOut& out, hipStream_t st, int* fallback_out, bool use_bootstrap, bool* bootstrap_overflow) {
  PagePlan plan;
  if (int rc = page_enqueue(idx, qw, n_qtiles, nq, nq_pad, page_k, bounded, out, st, use_bootstrap, &plan, nullptr)) return rc;
  PROQA_HIP(hipStreamSynchronize(st));
  return page_complete(idx, qw, n_qtiles, nq, nq_pad, page_k, bounded, out, st, plan, fallback_out, bootstrap_overflow);
}

// ---- large k in one pass -----------------------------------------------------------------------------------------
// Paging a large k (trec_process.py:76 asks for 10000, online_sampler.py:113 for 5000) costs a whole search per 4096
// results, each with ~15 rounds whose merges re-sort 8192 keys.  Instead: (1) the ordinary small-k rounds over a SAMPLE
// of the shard give every query the r-th best score t of n_s rows; the number of rows of the whole shard that beat t
// is then ~ r N / n_s, with a relative spread of 1/sqrt(r); n_s is chosen so that this is f k with f = 1 / (1 - 5.5 /
// sqrt(r)) -- fewer than k such rows is a 5.5-sigma event, more than the merge holds a 5-sigma one.  (2) ONE filter launch over all N rows against t, into
// lane lists spread over enough chunks (~8 records per list), and ONE merge per query that sorts the ~f k survivors.
// (3) The result is exact iff every query collected >= k rows (all rows above t are then known, the best k of them are
// the best k of the shard) and nothing overflowed; otherwise -- an ordered corpus the sample misjudges -- the search
// is repeated page by page.  The sample's slabs are spread over the shard, not taken from its head, for that reason.
struct OnePassPlan {
  bool use = false;
  int r = 0;                // rank whose score in the sample becomes the threshold
  long long n_sample = 0;   // sample rows
  int sort_cap = 0;         // keys the final merge holds
  unsigned want_chunks = 0;
  unsigned lane_cap = 0;       // records per lane list of the big launch
  bool compact = false;        // the big launch logs 8-byte keys (mips_filter_f16<COMPACT>)
  long long max_queries = 0;   // per launch, within kOnePassMaxStoreBytes of candidate store
  double expected = 0.0;       // rows expected to beat the sampled threshold, per query
};

const bool kOnePass = !(getenv("PROQA_ONE_PASS") && atoi(getenv("PROQA_ONE_PASS")) == 0);   // developer A/B switch
const bool kOnePassCompact = !(getenv("PROQA_ONE_PASS_COMPACT") && atoi(getenv("PROQA_ONE_PASS_COMPACT")) == 0);   // the same, for its compact lists
const bool kOnePassTwoStep = !(getenv("PROQA_ONE_PASS_TWO_STEP") && atoi(getenv("PROQA_ONE_PASS_TWO_STEP")) == 0);   // the same, for its sample
constexpr size_t kOnePassMaxStoreBytes = 24ull << 30;   // a batch whose store would be larger is searched in groups

OnePassPlan plan_one_pass(const proqa_index* idx, int64_t nq_pad, int k, bool latency_bound) {
  OnePassPlan p;
  // k <= 1024 (down to ~670: below that the rank-256 sample would have to cover more than a quarter of the shard) only
  // for small batches, where the launches of ~20 rounds cost more than a second look at 17-24 % of the shard (one
  // question, k = 1000, 18M rows: 1.65 -> 1.12 ms; at 2032 queries the rounds are as fast or faster)
  if (!kOnePass || idx->exact || k <= 512 || (k <= kPageK && !latency_bound) || idx->n < 65536) return p;
  // the smallest rank (cheapest sample) whose 5-sigma candidate count still fits the largest merge.  A small batch is
  // bound by the number of launches, not by the candidates: rank 256 takes the bootstrap and four rounds where rank 512
  // takes thirteen, for 15 % more candidates
  double expected = 0.0, most = 0.0;
  const int ranks[3] = {k <= 2560 || latency_bound ? 256 : 512, 512, 1024};
  for (int r : ranks) {
    const double sigma = 1.0 / std::sqrt((double)r);
    const double f = 1.0 / (1.0 - 5.5 * sigma);
    p.r = r;
    expected = f * k;
    most = expected * (1.0 + 5.0 * sigma);
    if (most <= kOnePassSortKeys) break;
  }
  if (most > kOnePassSortKeys) return p;                       // k beyond ~11700: pages
  p.sort_cap = most <= kMaxSortKeys ? kMaxSortKeys : (most <= kBigSortKeys ? kBigSortKeys : kOnePassSortKeys);
  p.n_sample = round_up<long long>((long long)(p.r * (double)idx->n / expected), kStageRows);
  if (p.n_sample > idx->n / 4) return p;                       // k is a large part of the shard
  // Dense regime (thousands of queries x a large k: more than every second 32-row unit of a wave holds a score above its
  // threshold): compact lists of 8-byte keys, ~24 per list of 70; otherwise column records, ~8 per list of 24.
  // (compact lists exist for the 8 x 64-query tiling only: with PROQA_FILTER_QW=4 / 1 the launch logs column records)
  p.compact = kOnePassCompact && !latency_bound && kWideQw == 2 && expected / (double)idx->n * 2048.0 > 0.5;
  p.lane_cap = p.compact ? (unsigned)kCompactLaneCap : (unsigned)kOnePassLaneCap;
  const double per_list = p.compact ? 24.0 : 8.0;
  p.want_chunks = round_up<unsigned>((unsigned)std::ceil(expected / (2.0 * per_list)), 8);   // two lists per chunk
  if ((long long)p.want_chunks * kStageRows > idx->n) return p;
  {
    // chunks are whole stages: a shard of few stages per chunk may end up with fewer chunks than asked for
    const long long rpc = round_up<long long>(ceil_div<long long>(idx->n, p.want_chunks), kStageRows);
    const long long chunks = ceil_div<long long>(idx->n, rpc);
    if (expected / (2.0 * chunks) > 1.5 * per_list) return p;  // the lists would spill (24 / 64 keys per list): pages
  }
  // queries one launch can take within the store budget (whole query tiles)
  const size_t per_query = (size_t)(p.want_chunks + 8) * 2 * p.lane_cap * sizeof(WaveRecord);
  size_t budget = kOnePassMaxStoreBytes;
  if (const char* v = getenv("PROQA_ONE_PASS_STORE_MB")) budget = (size_t)atoll(v) << 20;   // tests: force the grouping
  p.max_queries = (long long)(budget / per_query) / 512 * 512;
  if (p.max_queries < 512 && nq_pad > p.max_queries) return p;
  p.expected = expected;
  p.use = true;
  return p;
}

// steps (2) and (3) of search_one_pass: the thresholds stand, the sample's lists are forgotten; one launch over the shard,
// one merge, the result written optimistically before the one host sync
int one_pass_big_launch(proqa_index* idx, const OnePassPlan& pl, const RoundShape& shape, int qw, unsigned n_qtiles, int64_t nq,
                        int64_t nq_pad, int k, const PageOut& out, hipStream_t st, int sample_rounds, bool* done) {
  (void)pl;
  PROQA_HIP(hipMemsetAsync(idx->run_n, 0, (size_t)idx->ws_nq_pad * sizeof(unsigned), st));
  unsigned* word = idx->overflow + 1;   // [1]: the pass overflowed, [2]: a query came back short
  if (int rc = run_round(idx, Slab{0, idx->n}, qw, n_qtiles, (unsigned)nq_pad, k, false, false, false, word, st, nullptr, nullptr,
                         shape))
    return rc;
  PROQA_HIP(launch_flag_short_lists(idx->run_n, nq, (unsigned)k, word + 1, st));
  PROQA_HIP(launch_finalize(idx->run_keys, idx->run_n, nq, k, out.idx_offset, out.D, out.I, out.out_stride, 0, idx->overflow, nullptr,
                            idx->mirror, idx->stat_dev, st));
  PROQA_HIP(hipEventRecord(idx->ev[1], st));
  PROQA_HIP(hipStreamSynchronize(st));
  idx->stats.rounds += sample_rounds + 1;
  *done = !idx->mirror->overflow[1] && !idx->mirror->overflow[2];
  return PROQA_OK;
}

// The big launch of a one-pass search for a FEW queries (one question with k = 5000, online_sampler.py:113) on the
// int8 copy: such a launch is an HBM stream, and the int8 rows are half the bytes (18M rows: 0.68 -> ~0.37 ms).  The scan
// logs {row0, nominee bits} records exactly as a nominating round does; rescore_nominated_lists re-scores the nominated rows
// (~2.5 x the rows that pass: ~19 k x 256 B for k = 5000, nothing for a few queries -- for thousands it would be tens of GB,
// which is why the batched large-k search stays on the fp16 scan) and writes the keys that beat the sampled threshold as
// compact lists; the compact merge and everything behind it are the fp16 launch's.  *handled = false: not taken (no usable
// copy, a shape the lists do not fit) -- the caller runs the fp16 launch.
const bool kOnePassI8 = !(getenv("PROQA_ONE_PASS_I8") && atoi(getenv("PROQA_ONE_PASS_I8")) == 0);   // developer A/B switch
// When it pays: the launch saves 128 B per row of the shard and costs a 256-byte gather (plus a share of an MFMA and of a merge)
// per nominated row, ~3 x expected per query -- measured break-even near rows = 12 x queries x expected (18M rows, k = 5000:
// -32 % for one question, -26 % for 64, -19 % for 128, -7 % for 256; 2.25M rows: -15 % for one, 0 for 32); taken from 16 x.
// (developer override PROQA_ONE_PASS_I8_MAX_QUERIES: up to that many queries whatever the shard)
const int kOnePassI8MaxQueries = getenv("PROQA_ONE_PASS_I8_MAX_QUERIES") ? atoi(getenv("PROQA_ONE_PASS_I8_MAX_QUERIES")) : 0;
int one_pass_big_launch_i8(proqa_index* idx, const OnePassPlan& pl, int qw, unsigned n_qtiles, int64_t nq, int64_t nq_pad, int k,
                           const PageOut& out, hipStream_t st, int sample_rounds, bool* done, bool* handled) {
  *handled = false;
  if (!kOnePassI8 || qw != 1 || idx->exact || idx->nominate_mode == 0 || idx->n < kNominateMinRows) return PROQA_OK;
  if (kOnePassI8MaxQueries > 0 ? nq > kOnePassI8MaxQueries : 16.0 * (double)nq * pl.expected > (double)idx->n) return PROQA_OK;
  if (idx->q8_epoch != idx->rows_epoch) {
    if (idx->nominate_mode == 1 && idx->q8_short_lived_builds >= 2) return PROQA_OK;   // rows that keep changing: see setup_nominate
    if (int rc = ensure_q8(idx, st)) return rc;
  }
  if (!idx->q8_usable || (idx->q8_unprofitable && idx->nominate_mode != 2)) return PROQA_OK;
  // lists: ~8 records (of kNominateLaneCap = 32) per lane list of the scan at 3 nominated rows per passing row; ~24 keys (of
  // kCompactKeys = 64) per compact list
  const unsigned want = round_up<unsigned>((unsigned)std::ceil(3.0 * pl.expected / 16.0), 8);
  const LaunchGeom g = geometry(idx->n, n_qtiles, false, k, want);
  const unsigned in_lists = 2 * g.chunks;
  unsigned groups = std::max(2u, round_up<unsigned>((unsigned)std::ceil(pl.expected / 24.0), 2));
  const unsigned per_group = ceil_div<unsigned>(in_lists, groups);
  if (per_group == 0 || per_group > 64) return PROQA_OK;
  groups = round_up<unsigned>(ceil_div<unsigned>(in_lists, per_group), 2);
  const unsigned out_chunks = groups / 2, out_stride = round_up<unsigned>(out_chunks, 8);
  if (int rc = ensure_store(idx, round_up<unsigned>(g.chunks, 8), n_qtiles, nq_pad, kNominateLaneCap))
    return rc == PROQA_ENOMEM ? PROQA_OK : rc;
  {
    const size_t records = (size_t)out_chunks * nq_pad * 2 * kCompactLaneCap, counts = (size_t)nq_pad * out_stride * 2;
    if (records > idx->emit_records || counts > idx->emit_counts) {
      PROQA_HIP(hipStreamSynchronize(st));
      if (idx->emit_log) (void)hipFree(idx->emit_log);
      if (idx->emit_cnt) (void)hipFree(idx->emit_cnt);
      idx->emit_log = nullptr;
      idx->emit_cnt = nullptr;
      idx->emit_records = idx->emit_counts = 0;
      PROQA_HIP(hipMalloc((void**)&idx->emit_log, records * sizeof(WaveRecord)));
      PROQA_HIP(hipMalloc((void**)&idx->emit_cnt, counts * sizeof(unsigned)));
      idx->emit_records = records;
      idx->emit_counts = counts;
    }
  }
  *handled = true;
  // (the merge reads the list lengths of every padded query; the re-scoring writes those of the real ones)
  PROQA_HIP(hipMemsetAsync(idx->emit_cnt, 0, (size_t)nq_pad * out_stride * 2 * sizeof(unsigned), st));
  PROQA_HIP(launch_prep_queries_i8(idx->xq_pad, idx->ws_nq_pad, idx->col, idx->qstats, idx->xq8, idx->qparams, idx->stat_nom, st));
  PROQA_HIP(hipMemsetAsync(idx->run_n, 0, (size_t)idx->ws_nq_pad * sizeof(unsigned), st));
  unsigned* word = idx->overflow + 1;   // [1]: the pass overflowed, [2]: a query came back short
  FilterArgsI8 fa;
  fa.xq8 = idx->xq8;
  fa.xb8 = idx->xb8;
  fa.slab_row0 = 0;
  fa.slab_row1 = idx->n;
  fa.rows_per_chunk = g.rows_per_chunk;
  fa.tau = idx->tau;
  fa.qp = idx->qparams;
  fa.blk = idx->blk8;
  fa.store = store_of(idx, (unsigned)nq_pad, n_qtiles, kNominateLaneCap, round_up<unsigned>(g.chunks, 8));
  fa.overflow = word;
  fa.flags = kFilterFlags;
  // (row-split launch as in run_round: the 1 / 2 / 4 query blocks that hold queries, the eight waves share the units of the stream)
  fa.q_blocks = kRowSplit && nq <= 128 ? (nq <= 32 ? 1u : (nq <= 64 ? 2u : 4u)) : 0u;
  PROQA_HIP(launch_filter_i8(fa, qw, g.grid, st, false));
  CandidateStore emitted;
  emitted.lane_log = idx->emit_log;
  emitted.lane_cnt = idx->emit_cnt;
  emitted.spill_log = idx->spill_log;   // (the merge reads a spill counter per chunk up front whatever the lists hold; compact
  emitted.spill_cnt = idx->spill_cnt;   //  lists have no spill log and the value is not used: any valid counters do)
  emitted.nq_pad = (unsigned)nq_pad;
  emitted.n_qtiles = n_qtiles;
  emitted.lane_cap = (unsigned)kCompactLaneCap;
  emitted.n_chunks = out_stride;
  RescoreArgs ra;
  ra.in = fa.store;
  ra.in_lists = in_lists;
  ra.lists_per_group = per_group;
  ra.out = emitted;
  ra.xq16 = idx->xq_pad;
  ra.xb16 = idx->xb;
  ra.tau = idx->tau;
  ra.overflow = word;
  ra.stat_nominated = idx->stat_nom;
  PROQA_HIP(launch_rescore_nominated_lists(ra, groups, (unsigned)nq, st));
  MergeArgs ma = {};
  ma.store = emitted;
  ma.n_chunks = out_chunks;
  ma.qw = (unsigned)qw;
  ma.run_keys = idx->run_keys;
  ma.run_n = idx->run_n;
  ma.tau = idx->tau;
  ma.k = k;
  ma.sort_cap = pl.sort_cap;
  ma.stat_candidates = idx->stat_dev;
  ma.overflow = word;
  ma.compact = 1;
  PROQA_HIP(launch_merge(ma, (unsigned)nq_pad, st));
  PROQA_HIP(launch_flag_short_lists(idx->run_n, nq, (unsigned)k, word + 1, st));
  PROQA_HIP(launch_finalize(idx->run_keys, idx->run_n, nq, k, out.idx_offset, out.D, out.I, out.out_stride, 0, idx->overflow, nullptr,
                            idx->mirror, idx->stat_dev, st, idx->stat_nom));
  PROQA_HIP(hipEventRecord(idx->ev[1], st));
  PROQA_HIP(hipStreamSynchronize(st));
  idx->stats.rounds += sample_rounds + 1;
  idx->stats.nomination = 1;
  idx->stats.nominated = (int64_t)idx->mirror->nominated;
  *done = !idx->mirror->overflow[1] && !idx->mirror->overflow[2];
  return PROQA_OK;
}

// returns PROQA_OK with *done = false when the estimate failed (the caller searches page by page)
int search_one_pass(proqa_index* idx, const OnePassPlan& pl, int qw, unsigned n_qtiles, int64_t nq, int64_t nq_pad, int k,
                    const PageOut& out, hipStream_t st, bool* done) {
  *done = false;
  const RoundShape shape{pl.want_chunks, pl.lane_cap, pl.sort_cap, pl.compact};
  // the big launch's store first, so that the sample rounds do not allocate a small one that is thrown away
  {
    const LaunchGeom g = geometry(idx->n, n_qtiles, false, k, shape.want_chunks);
    if (int rc = ensure_store(idx, round_up<unsigned>(g.chunks, 8), n_qtiles, nq_pad, shape.lane_cap))
      return rc == PROQA_ENOMEM ? PROQA_OK : rc;   // no room for the deep lists beside the caller's tensors: page by page
  }
  // The compact lists of the dense launch hold 64 keys each and are sized for hits spread evenly over the shard.  A corpus
  // ordered by document puts a query's hits into a few chunks: a list wraps, the launch is void.  Before going page by page
  // (several times slower) the launch is repeated ONCE over four times the chunks -- against the thresholds the first
  // attempt's merge left, which are still lower bounds of the k-th best scores and usually tighter.
  auto big_launch_with_retry = [&](int sample_rounds) -> int {
    {   // a few queries: the launch over the shard on the int8 copy (a void attempt -- its lists overflowed -- goes to pages)
      bool handled = false;
      if (int rc8 = one_pass_big_launch_i8(idx, pl, qw, n_qtiles, nq, nq_pad, k, out, st, sample_rounds, done, &handled)) return rc8;
      if (handled) return PROQA_OK;
    }
    int rc = one_pass_big_launch(idx, pl, shape, qw, n_qtiles, nq, nq_pad, k, out, st, sample_rounds, done);
    if (rc || *done || !shape.compact || !idx->mirror->overflow[1]) return rc;
    RoundShape wide = shape;
    wide.want_chunks = round_up<unsigned>((unsigned)std::min<long long>(4ll * shape.want_chunks, idx->n / kStageRows), 8);
    if (wide.want_chunks <= shape.want_chunks) return rc;
    const LaunchGeom g = geometry(idx->n, n_qtiles, false, k, wide.want_chunks);
    if (ensure_store(idx, round_up<unsigned>(g.chunks, 8), n_qtiles, nq_pad, wide.lane_cap) != PROQA_OK) return PROQA_OK;   // no room: pages
    PROQA_HIP(hipMemsetAsync(idx->overflow + 1, 0, 2 * sizeof(unsigned), st));
    return one_pass_big_launch(idx, pl, wide, qw, n_qtiles, nq, nq_pad, k, out, st, sample_rounds, done);
  };
  // (1) thresholds from the sample (the round words were zeroed by prep_queries).  Overflow in here is harmless: it loosens the estimate.
  const int r = pl.r;
  // A rank beyond the bootstrap's (r = 512, 1024: thousands of queries with k in the thousands) would need ~7 rounds of
  // r (g - 1) candidates each to reach the r-th best of the sample (6.8 k candidates per query for k = 10000: 4.5 of the
  // 29 ms of 6980 queries over 8.8M rows).  Two steps instead: the bootstrap's rank-r' score t' (r' ~ 64 of 8192 rows)
  // is a coarse estimate of the same quantile; ONE round over the rest of the sample against t' logs c = r' m / 8192
  // rows per query (~2 k), with a spread of 1 / sqrt(r'), and its merge keeps the best r of them: the r-th best of the
  // sample, exactly, unless fewer than r - r' rows beat t' (below; the big launch then overflows and the search goes
  // page by page) or more than the merge holds (the estimate loosens).
  if (r > kBootstrapMaxK && kOnePassTwoStep && idx->bootstrap_rows > 0 && pl.n_sample >= 8ll * kBootstrapMaxRows) {
    const long long boot_rows = kBootstrapMaxRows;
    const long long m = pl.n_sample - boot_rows;
    // Fewer than r - r' of the m sample rows beat the r'-th best of the bootstrap's 8192 iff the (r - r')-th best of those
    // m rows is among the best r' of all m + 8192: with lambda = (r - r') 8192 / m head rows expected above that score,
    // a Poisson upper tail P(X >= r').  The smallest r' that puts it below 1e-12 per query.
    auto poisson_tail = [](double lambda, int at_least) {
      double term = std::exp(-lambda), below = 0.0;   // P(X = 0)
      for (int i = 0; i < at_least; ++i) {
        below += term;
        term *= lambda / (double)(i + 1);
      }
      double tail = 0.0;                                // sum the tail itself: 1 - below loses it to rounding
      for (int i = at_least; i < at_least + 200; ++i) {
        tail += term;
        term *= lambda / (double)(i + 1);
      }
      (void)below;
      return tail;
    };
    int r1 = 0;
    double c = 0.0;
    for (int cand = 32; cand <= kBootstrapMaxK; cand += 16) {
      c = (double)cand * (double)m / (double)boot_rows;
      if (poisson_tail((double)(r - cand) * (double)boot_rows / (double)m, cand) < 1e-12) {
        r1 = cand;
        break;
      }
    }
    if (r1 && c * (1.0 + 5.0 / std::sqrt((double)r1)) + r1 <= (double)kBigSortKeys) {
      if (int rc = run_bootstrap(idx, boot_rows, (unsigned)nq_pad, r1, st, r)) return rc;
      // the sample slab: the middle of the shard (the bootstrap looked at its head)
      long long r0 = std::max<long long>(boot_rows, ((idx->n - m) / 2) / kStageRows * kStageRows);
      if (r0 + m > idx->n) r0 = idx->n - m;
      RoundShape sample_shape;
      sample_shape.want_chunks = round_up<unsigned>((unsigned)std::ceil(c / 8.0), 8);   // ~4 records per lane list
      sample_shape.lane_cap = (unsigned)kOnePassLaneCap;
      sample_shape.sort_cap = c * (1.0 + 5.0 / std::sqrt((double)r1)) + r1 <= (double)kMidSortKeys ? kMidSortKeys : kBigSortKeys;
      if (int rc = run_round(idx, Slab{r0, r0 + m}, qw, n_qtiles, (unsigned)nq_pad, r, false, false, false, idx->overflow, st,
                             nullptr, nullptr, sample_shape))
        return rc;
      if (kDebugCand) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> per_query((size_t)nq);
        (void)hipMemcpy(per_query.data(), idx->stat_dev, (size_t)nq * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int64_t q = 0; q < nq; ++q) tot += per_query[q];
        fprintf(stderr, "two-step sample: bootstrap rank %d of %lld rows, then rows [%lld, %lld) for rank %d: %.0f candidates per query "
                "(expected %.0f)\n", r1, boot_rows, r0, r0 + m, r, (double)tot / (double)nq, c);
      }
      return big_launch_with_retry(2);
    }
  }
  long long boot = 0;
  if (idx->bootstrap_rows > 0 && r <= kBootstrapMaxK && r <= idx->bootstrap_rows / 4 && pl.n_sample >= 4ll * idx->bootstrap_rows)
    boot = std::min<long long>(idx->bootstrap_rows, kBootstrapMaxRows);
  const int first_cap = (sort_capacity(r) - r) / kStageRows * kStageRows;
  // the sample rounds run on this search's deep lane lists (24 records): their candidate budget is what one merge holds
  // next to the r running keys, less 15 % for the spread, not the 640 of a search on 8-record lists
  const double g_cap = idx->growth > 0 ? idx->growth : (qw == 1 ? 8 : 4);
  const double growth = std::min(g_cap, 0.85 * (sort_capacity(r) - r) / r);
  std::vector<Slab> slabs = plan_slabs(pl.n_sample, std::min<int>(idx->first_slab_rows, first_cap), growth, boot);
  // Leaping sample rounds (plan_leap; r <= 256: the merge's early exit reads the whole running list from one key per thread):
  // thresholds at a rank j < r, two rounds instead of three for one question with k = 5000.  Nothing verifies them and nothing
  // needs to: a round that falls short leaves the sample's r-th best too LOW, i.e. a looser threshold for the launch over
  // the shard, whose result is checked as ever (>= k rows per query, no overflow).
  int leap_rank_sample = 0;
  if (boot && r <= 256 && idx->leap_mode && env_int("PROQA_LEAP", 1) != 0 && idx->growth == 0 && kGrowthList.empty()) {
    // (the plan is kept: the negative-binomial sums for r = 256 are ~0.2 ms of host time)
    const long long key[4] = {pl.n_sample, boot, r, qw};
    if (std::memcmp(key, idx->leap_sample_key, sizeof key) != 0) {
      std::memcpy(idx->leap_sample_key, key, sizeof key);
      const LeapPlan fresh = plan_leap(pl.n_sample, boot, r, qw, false, 256);
      idx->leap_sample_rounds = fresh.rounds;
      idx->leap_sample_rank = fresh.rank;
    }
    LeapPlan lp;
    lp.rounds = idx->leap_sample_rounds;
    lp.rank = idx->leap_sample_rank;
    if (lp.rounds && lp.rounds < (int)slabs.size()) {
      const double rho = std::pow((double)pl.n_sample / (double)boot, 1.0 / lp.rounds);
      slabs = plan_slabs_equal(pl.n_sample, boot, rho - 1.0 + 1e-6);
      leap_rank_sample = lp.rank;
    }
  }
  if ((int)slabs.size() + 4 > kMaxRounds) return PROQA_OK;
  if (boot)
    if (int rc = run_bootstrap(idx, boot, (unsigned)nq_pad, r, st, 0, leap_rank_sample)) return rc;
  // spread the slabs over [boot, n): slab i keeps its length and starts i/m of the way through
  {
    const size_t m = slabs.size();
    long long prev_end = boot;
    bool fits = true;
    std::vector<Slab> spread(m);
    for (size_t i = 0; i < m; ++i) {
      const long long len = slabs[i].r1 - slabs[i].r0;
      long long r0 = boot + (long long)((double)i / (double)m * (double)(idx->n - boot)) / kStageRows * kStageRows;
      r0 = std::max(r0, prev_end);
      if (r0 + len > idx->n) fits = false;
      spread[i] = {r0, r0 + len};
      prev_end = r0 + len;
    }
    if (fits) slabs.swap(spread);
  }
  long long seen = boot;
  for (size_t i = 0; i < slabs.size(); ++i) {
    // while fewer than r rows have been merged the threshold is still -inf: every row is logged
    idx->round_next_rank = leap_rank_sample && i + 1 < slabs.size() ? leap_rank_sample : 0;
    const int rc_round = run_round(idx, slabs[i], qw, n_qtiles, (unsigned)nq_pad, r, false, seen < r, false, idx->overflow, st, nullptr,
                                   nullptr);
    idx->round_next_rank = 0;
    if (int rc = rc_round)
      return rc;
    seen += slabs[i].r1 - slabs[i].r0;
    if (kDebugCand) {
      (void)hipStreamSynchronize(st);
      std::vector<unsigned long long> per_query((size_t)nq);
      (void)hipMemcpy(per_query.data(), idx->stat_dev, (size_t)nq * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long c = 0;
      for (int64_t q = 0; q < nq; ++q) c += per_query[q];
      unsigned ov = 0;
      (void)hipMemcpy(&ov, idx->overflow, sizeof ov, hipMemcpyDeviceToHost);
      fprintf(stderr, "sample round %zu: rows [%lld, %lld), rank %d, cumulative candidates %llu (%.0f per query), overflow word %u\n", i,
              slabs[i].r0, slabs[i].r1, r, c, (double)c / (double)nq, ov);
    }
  }
  return big_launch_with_retry((int)slabs.size());
}

// after the host sync of a search whose rounds ran on the int8 copy: statistics, and the profitability check -- a corpus
// whose scores the eight bits cannot separate (nominations per query and row far above what pays for the halved scan) goes
// back to the fp16 scan until its rows change.  Exactness never depends on this: every nominated row is re-scored.
constexpr int kProbeAfterFirst = 8, kProbeAfterMost = 64;
int nomination_state_of(const proqa_index* idx) {   // proqa_search_stats::nomination_state
  if (idx->nominate_mode == 0 || idx->exact) return 0;
  if (idx->q8_epoch == idx->rows_epoch && (!idx->q8_usable || (idx->q8_unprofitable && idx->nominate_mode != 2))) return 2;
  return 1;
}
void note_nomination(proqa_index* idx, int64_t nq) {
  idx->stats.nomination_state = nomination_state_of(idx);
  if (!idx->q8_active) return;
  idx->q8_active = false;
  idx->stats.nomination = 1;
  idx->stats.nominated = (int64_t)idx->mirror->nominated;
  const double per_query = (double)idx->stats.nominated / (double)std::max<int64_t>(nq, 1);
  const double limit = std::max(4096.0, (double)idx->n / 2048.0);
  // (overflow bit 3 alone: a leaping round fell short -- note_leap's business, not a property of the int8 copy)
  const bool bad = per_query > limit || (idx->stats.fallback_rounds > 0 && (idx->overflow_bits & ~8u) != 0);
  if (bad && per_query <= limit && idx->used_small_merge && idx->small_merge_ok && (idx->overflow_bits & ~8u) == 2u) {
    // what overflowed was the capacity of the 1024-key merge alone (no lane list of the scan: bit 0), and the nominations
    // are within the limit: these rows nominate more per round than that merge is sized for -- the 2048-key merge from now
    // on, no suspension (if that overflows too, the next search suspends the rounds)
    idx->small_merge_ok = false;
    log_line("index %p: a round's nominations overflowed the 1024-key merge (%.0f rows re-scored per query, %d overflow-safe rounds): "
             "2048-key merges from now on", (void*)idx, per_query, idx->stats.fallback_rounds);
    idx->stats.nomination_state = nomination_state_of(idx);
    return;
  }
  if (bad) {
    // One such batch may be an outlier (an adversarial or degenerate set of queries) on rows that quantise well: the
    // suspension is lifted by a later search that passes.  First probe after 8 eligible searches, then 16, 32, 64, 64, ...
    idx->q8_probe_after = idx->q8_unprofitable ? std::min(2 * std::max(idx->q8_probe_after, kProbeAfterFirst), kProbeAfterMost)
                                               : kProbeAfterFirst;
    idx->q8_suspended_searches = 0;
    log_line("index %p: int8 nomination scan %s (%.0f rows re-scored per query against a limit of %.0f, %d overflow-safe rounds): "
             "fp16 scan for the next %d eligible searches", (void*)idx, idx->q8_unprofitable ? "stays suspended" : "suspended", per_query,
             limit, idx->stats.fallback_rounds, idx->q8_probe_after - 1);
    idx->q8_unprofitable = true;
  } else if (idx->q8_unprofitable && idx->nominate_mode != 2) {
    log_line("index %p: int8 nomination scan resumed (%.0f rows re-scored per query)", (void*)idx, per_query);
    idx->q8_unprofitable = false;
    idx->q8_suspended_searches = 0;
    idx->q8_probe_after = 0;
  }
  idx->stats.nomination_state = nomination_state_of(idx);
}

// after the host sync of a search that leapt: a round that fell short pauses the leaps of this index
int leap_state_of(const proqa_index* idx) {   // proqa_search_stats::leap_state
  if (!idx->leap_mode || idx->exact) return 0;
  return idx->leap_skip > 0 && idx->leap_epoch == idx->rows_epoch ? 2 : 1;
}
void note_leap(proqa_index* idx) {
  idx->stats.leap_state = leap_state_of(idx);
  if (!idx->leap_active) return;
  idx->leap_active = false;
  if (idx->overflow_bits) {
    // bit 3: a round fell short; any other: scores that tie in numbers (the rows a leap logs tie with more).  What it cost
    // decides how soon the leaps pause: up to 256 short queries searched again by themselves (idx->rescue) are ~10 % of a
    // search -- three strikes each, one taken back by every clean leap, a pause at eight: leaps go on while fewer than a
    // quarter of the searches fall short (they pay up to a half); more short queries (up to a whole second search) or the
    // slab re-scan for every query -- eight strikes, a pause at once
    const bool cheap = !idx->rescue.ids.empty() && idx->rescue.ids.size() <= (size_t)kRescueCheap;
    idx->leap_strikes = std::min(idx->leap_strikes + (cheap ? 3 : 8), 16);
    if (idx->leap_strikes >= 8) {
      idx->leap_pause = idx->leap_pause ? std::min(2 * idx->leap_pause, 1024) : 16;
      idx->leap_skip = idx->leap_pause;
      idx->leap_strikes = 4;
      log_line("index %p: a leaping round found fewer than k rows above its threshold or overflowed (%d overflow-safe rounds, overflow "
               "bits %u): ordinary rounds for the next %d searches", (void*)idx, idx->stats.fallback_rounds, idx->overflow_bits,
               idx->leap_skip);
    } else {
      log_line("index %p: a leaping round found fewer than k rows above its threshold for %zu quer%s: leaps go on (%d strikes of 8)",
               (void*)idx, idx->rescue.ids.size(), idx->rescue.ids.size() == 1 ? "y" : "ies", idx->leap_strikes);
    }
  } else {
    if (idx->leap_strikes > 0) --idx->leap_strikes;
    if (idx->leap_pause) {
      log_line("index %p: leaping rounds resumed", (void*)idx);
      idx->leap_pause = 0;
    }
  }
  idx->stats.leap_state = leap_state_of(idx);
}

int finish_pending(proqa_index* idx, int* rewritten);
int search_device_impl(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset, float* D_dev,
                       int64_t* I_dev, hipStream_t st, bool defer, uint32_t* status_dev);

// The queries page_complete listed in idx->rescue (leaping rounds left them short; every other query's result is verified):
// their padded fp16 rows are gathered into a batch of their own, searched on ordinary rounds with this handle's workspace
// (the search that owned it is complete), and their result rows written over the optimistic ones.  Waits for the stream.
int rescue_short_queries(proqa_index* idx) {
  proqa_index::Rescue rq;
  std::swap(rq, idx->rescue);
  const int s = (int)rq.ids.size(), k = rq.k;
  if (s == 0) return PROQA_OK;
  const size_t off_xq = round_up<size_t>((size_t)s * sizeof(int), 256), off_d = off_xq + (size_t)s * kDim * 2,
               off_i = round_up<size_t>(off_d + (size_t)s * k * sizeof(float), 256), bytes = off_i + (size_t)s * k * sizeof(long long);
  if (bytes > idx->rescue_bytes) {
    if (idx->rescue_buf) (void)hipFree(idx->rescue_buf);
  if (idx->emit_log) (void)hipFree(idx->emit_log);
  if (idx->emit_cnt) (void)hipFree(idx->emit_cnt);
    idx->rescue_buf = nullptr;
    idx->rescue_bytes = 0;
    PROQA_HIP(hipMalloc(&idx->rescue_buf, bytes));
    idx->rescue_bytes = bytes;
  }
  char* base = (char*)idx->rescue_buf;
  int* ids_dev = (int*)base;
  float* D_tmp = (float*)(base + off_d);
  long long* I_tmp = (long long*)(base + off_i);
  PROQA_HIP(hipMemcpyAsync(ids_dev, rq.ids.data(), (size_t)s * sizeof(int), hipMemcpyHostToDevice, rq.st));
  PROQA_HIP(launch_gather_query_rows(idx->xq_pad, ids_dev, s, base + off_xq, rq.st));
  const proqa_search_stats outer = idx->stats;
  const int mode = idx->leap_mode;
  idx->leap_mode = 0;   // (ordinary rounds; the pause this search earned stays as note_leap left it)
  // the second search is part of the caller's ONE search: it does not count towards the automatic mode of the int8 scan
  // (a suspended scan's probe distance, the copy's searches), whose state is put back as the caller's search left it
  const bool unprofitable = idx->q8_unprofitable, small_ok = idx->small_merge_ok;
  const int suspended = idx->q8_suspended_searches, probe_after = idx->q8_probe_after, on_copy = idx->q8_searches_on_copy;
  const int rc = search_device_impl(idx, base + off_xq, s, PROQA_F16, k, rq.idx_offset, D_tmp, (int64_t*)I_tmp, rq.st, false, nullptr);
  idx->leap_mode = mode;
  idx->q8_unprofitable = unprofitable;
  idx->small_merge_ok = small_ok;
  idx->q8_suspended_searches = suspended;
  idx->q8_probe_after = probe_after;
  idx->q8_searches_on_copy = on_copy;
  if (rc) return rc;
  PROQA_HIP(launch_scatter_result_rows(D_tmp, I_tmp, ids_dev, s, k, rq.D, rq.I, rq.out_stride, rq.st));
  PROQA_HIP(hipStreamSynchronize(rq.st));
  const proqa_search_stats inner = idx->stats;
  idx->stats = outer;
  idx->stats.fallback_rounds += inner.rounds + inner.fallback_rounds;   // rounds run again
  idx->stats.candidates += inner.candidates;
  idx->stats.nominated += inner.nominated;
  idx->stats.total_ms += inner.total_ms;
  idx->stats.nomination_state = nomination_state_of(idx);
  idx->stats.leap_state = leap_state_of(idx);
  log_line("index %p: %d quer%s a leaping round left short searched again on ordinary rounds (%d rounds, %.3f ms)", (void*)idx, s,
           s == 1 ? "y" : "ies", inner.rounds, inner.total_ms);
  return PROQA_OK;
}

int search_device_impl_body(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset, float* D_dev,
                            int64_t* I_dev, hipStream_t st, bool defer, uint32_t* status_dev);

// `defer`: a search of the one-page kind is only ENQUEUED (idx->pending describes it; finish_pending completes it);
// every other kind runs to completion here.  `status_dev` (optional device word, written on the stream): 1 if the
// completion will rewrite the result, else 0.
int search_device(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset,
                  float* D_dev, int64_t* I_dev, hipStream_t st, bool defer = false, uint32_t* status_dev = nullptr) {
  const int rc = search_device_impl(idx, xq_dev, nq, dtype, k, idx_offset, D_dev, I_dev, st, defer, status_dev);
#ifdef PROQA_FILTER_STAMPS
  {
    unsigned long long h[8];
    read_filter_stamps(h);
    if (h[4])
      fprintf(stderr, "filter stamps (one wave per workgroup, %llu workgroups over all launches of the search): per unit MFMA section %.1f "
              "ticks, test section %.1f ticks; units per wave %.0f, wave lifetime per unit %.1f ticks; int8 scan: barrier + DMA issue %.1f "
              "ticks per unit, %.3f excursions per unit of %.1f ticks each\n", h[4], (double)h[0] / h[2],
              (double)h[1] / h[2], (double)h[2] / h[4], (double)h[3] / h[2], (double)h[5] / h[2], (double)h[7] / h[2],
              h[7] ? (double)h[6] / h[7] : 0.0);
  }
#endif
  // a search that ran to completion has a final result: its status word is 0 (the deferred kind writes the word itself)
  if (rc == PROQA_OK && status_dev && !idx->pending.active) PROQA_HIP(hipMemsetAsync(status_dev, 0, sizeof(uint32_t), st));
  return rc;
}

int search_device_impl(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset,
                       float* D_dev, int64_t* I_dev, hipStream_t st, bool defer, uint32_t* status_dev) {
  idx->rescue.ids.clear();
  if (int rc = search_device_impl_body(idx, xq_dev, nq, dtype, k, idx_offset, D_dev, I_dev, st, defer, status_dev)) return rc;
  return idx->rescue.ids.empty() ? PROQA_OK : rescue_short_queries(idx);
}

int search_device_impl_body(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset,
                            float* D_dev, int64_t* I_dev, hipStream_t st, bool defer, uint32_t* status_dev) {
  if (idx->pending.active)   // a begun search nobody finished (an error path of the caller): complete it, drop its result
    if (int rc = finish_pending(idx, nullptr)) return rc;
  if (nq < 0 || k <= 0) return fail(PROQA_EINVAL, "search: nq=%lld k=%d", (long long)nq, k);
  if (dtype != PROQA_F16 && dtype != PROQA_F32) return fail(PROQA_EINVAL, "search: bad dtype %d", dtype);
  if (idx->n >= (1ll << 32)) return fail(PROQA_EINVAL, "search: shard has >= 2^32 rows");
  idx->stats = {};
  idx->stats.nomination_state = nomination_state_of(idx);
  idx->stats.leap_state = leap_state_of(idx);
  idx->q8_active = false;
  idx->pending_nq = nq;
  idx->used_small_merge = false;
  idx->overflow_bits = 0;
  idx->leap_active = false;
  if (nq == 0) return PROQA_OK;
  PROQA_ON_DEVICE(idx->device);

  // wave tile: 2 query blocks of 32 per wave (512 queries per workgroup) unless the batch is small
  const int qw = nq > 256 ? kWideQw : 1;
  const unsigned tile_q = filter_tile_queries(qw);
  const unsigned n_qtiles = (unsigned)ceil_div<int64_t>(nq, tile_q);
  const int64_t nq_pad = (int64_t)n_qtiles * tile_q;
  const int page_size = k <= kPageK ? kPageK : kBigPageK;
  OnePassPlan one_pass = plan_one_pass(idx, nq_pad, k, qw == 1);
  if (one_pass.use && nq_pad > one_pass.max_queries) {
    // thousands of queries x a large k: in groups whose candidate store stays within budget (each group is a whole
    // search of its own; the corpus is small next to the scores, re-reading it per group costs nothing)
    const size_t esize = dtype == PROQA_F16 ? 2 : 4;
    proqa_search_stats sum = {};
    const int64_t n_groups = ceil_div<int64_t>(nq, one_pass.max_queries);
    const int64_t per_group = round_up<int64_t>(ceil_div<int64_t>(nq, n_groups), 512);   // even groups of whole tiles
    for (int64_t q0 = 0; q0 < nq; q0 += per_group) {
      const int64_t m = std::min<int64_t>(per_group, nq - q0);
      if (int rc = search_device(idx, (const char*)xq_dev + (size_t)q0 * kDim * esize, m, dtype, k, idx_offset,
                                 D_dev + (size_t)q0 * k, I_dev + (size_t)q0 * k, st))
        return rc;
      sum.rounds += idx->stats.rounds;
      sum.fallback_rounds += idx->stats.fallback_rounds;
      sum.candidates += idx->stats.candidates;
      sum.filter_ms += idx->stats.filter_ms;
      sum.total_ms += idx->stats.total_ms;
    }
    idx->stats = sum;
    return PROQA_OK;
  }
  if (int rc = ensure_workspace(idx, nq_pad, one_pass.use ? k : std::min(k, page_size))) return rc;

  PROQA_HIP(hipEventRecord(idx->ev[0], st));
  int fallback = 0;
  // pad the queries and reset the per-query state for the first (or only) page; float32 queries fp16 cannot hold
  // switch the index to exact-float32 mode unless rounding is allowed
  auto prep_first = [&]() -> int {
    const bool check_q = dtype == PROQA_F32;
    if (check_q) PROQA_HIP(hipMemsetAsync(idx->inexact, 0, 2 * sizeof(unsigned), st));
    PROQA_HIP(launch_prep_queries(xq_dev, dtype, nq, idx->ws_nq_pad, idx->xq_pad, idx->tau, idx->run_n, idx->stat_dev,
                                  nullptr, true, check_q ? idx->inexact : nullptr, idx->overflow, st, idx->short_rounds));
    if (check_q) {
      unsigned bad = 0;
      if (int rc = read_inexact(idx, "index_search (queries)", st, &bad)) return rc;
      if (bad && !idx->allow_rounding)
        if (int rc = enable_exact(idx, st)) return rc;
    }
    return PROQA_OK;
  };
  // The rounds of a one-page search of an fp16 index scan the int8 copy of the rows (nomination + exact re-scoring, see
  // mips_kernels.hip) when the batch is MFMA-bound, k small enough for a round's nominations to fit its merge and the
  // shard large enough to matter.  Called after prep_first (it reads the padded fp16 queries).
  auto setup_nominate = [&]() -> int {
    // Batches of <= 256 queries (one query block per wave: the HBM-bound regime) take it too -- the int8 rows are half the
    // bytes of the stream (one question over 18M rows: 0.89 -> 0.65 ms, 256 queries 1.48 -> 1.01 ms).
    // (mode 2, "always", also takes small shards: tests and experiments)
    if (!nomination_eligible(idx, k)) return PROQA_OK;
    if (idx->q8_epoch != idx->rows_epoch) {
      // rows that keep changing between searches (see q8_short_lived_builds): the first search after a change scans the fp16
      // rows, the copy is rebuilt once a second search finds the same rows (mode "always" rebuilds at once)
      if (idx->q8_searches_on_copy > 1) idx->q8_short_lived_builds = 0;   // the stale copy earned its build
      if (idx->nominate_mode == 1 && idx->q8_short_lived_builds >= 2 && idx->q8_seen_epoch != idx->rows_epoch) {
        idx->q8_seen_epoch = idx->rows_epoch;
        log_line("index %p: rows changed again after an int8 copy that served %d search(es): this search scans the fp16 rows, the "
                 "copy is rebuilt when the rows stay", (void*)idx, idx->q8_searches_on_copy);
        return PROQA_OK;
      }
      idx->q8_seen_epoch = idx->rows_epoch;
      // the int8 copy is missing or stale.  A search that runs to completion in this call builds it here (allocation, two
      // passes over the rows, one host read).  An ENQUEUED search (_begin) must not synchronise or allocate: it scans
      // the fp16 rows and leaves the build to its _finish, i.e. to the host wait the caller pays anyway
      // (proqa_index_prepare does the same ahead of time).
      if (defer) {
        idx->q8_build_due = true;
        return PROQA_OK;
      }
      if (int rc = ensure_q8(idx, st)) return rc;
    }
    if (!idx->q8_usable) return PROQA_OK;
    if (idx->q8_unprofitable && idx->nominate_mode != 2) {
      // suspended (note_nomination): every q8_probe_after-th eligible search tries the int8 rounds again
      if (++idx->q8_suspended_searches < idx->q8_probe_after) return PROQA_OK;
      log_line("index %p: re-probing the int8 nomination scan after %d searches on the fp16 rows", (void*)idx,
               idx->q8_suspended_searches);
    }
    PROQA_HIP(launch_prep_queries_i8(idx->xq_pad, idx->ws_nq_pad, idx->col, idx->qstats, idx->xq8, idx->qparams, idx->stat_nom, st));
    idx->q8_active = true;
    ++idx->q8_searches_on_copy;
    return PROQA_OK;
  };
  // ~670 <= k <= ~11700 on a shard much larger than k: one pass against sampled thresholds (search_one_pass)
  if (one_pass.use) {
    if (int rc = prep_first()) return rc;
    bool done = false;
    if (!idx->exact)
      if (int rc = search_one_pass(idx, one_pass, qw, n_qtiles, nq, nq_pad, k, PageOut{D_dev, (long long*)I_dev, idx_offset, k, 0},
                                   st, &done))
        return rc;
    if (done) {
      idx->stats.candidates += (int64_t)idx->mirror->candidates;
      (void)hipEventElapsedTime(&idx->stats.total_ms, idx->ev[0], idx->ev[1]);
      return PROQA_OK;
    }
    if (!idx->exact) ++fallback;   // the estimate failed: page by page from scratch
  }
  // k <= kPageK: one page.  Larger k that did not go (or get) through the one-pass search is served page by
  // page: page p re-runs the search restricted to keys strictly below the last key of page p-1.
  const int n_pages = ceil_div<int>(k, page_size);
  if (defer && n_pages == 1 && !fallback) {
    // the common search (k <= kPageK): everything goes on the stream, nothing is waited for
    if (int rc = prep_first()) return rc;
    if (idx->exact)
      PROQA_HIP(launch_query_margins(xq_dev, dtype, nq, idx->ws_nq_pad, idx->norm_stats, idx->xq32, idx->margin,
                                     idx->tau, idx->tau_filter, st));
    if (int rc = setup_nominate()) return rc;
    PagePlan plan;
    if (int rc = page_enqueue(idx, qw, n_qtiles, nq, nq_pad, k, false, PageOut{D_dev, (long long*)I_dev, idx_offset, k, 0}, st, true,
                              &plan, status_dev))
      return rc;
    proqa_index::Pending& pe = idx->pending;
    pe.active = true;
    pe.qw = qw;
    pe.n_qtiles = n_qtiles;
    pe.nq = nq;
    pe.nq_pad = nq_pad;
    pe.k = k;
    pe.dtype = dtype;
    pe.xq_dev = xq_dev;
    pe.D = D_dev;
    pe.I = (long long*)I_dev;
    pe.idx_offset = idx_offset;
    pe.st = st;
    pe.slabs.swap(plan.slabs);
    pe.boot = plan.boot;
    return PROQA_OK;
  }
  if (n_pages > 1) PROQA_HIP(hipMemsetAsync(idx->done, 0, (size_t)idx->ws_nq_pad, st));
  for (int p = 0; p < n_pages; ++p) {
    const int page_k = std::min(page_size, k - p * page_size);
    if (p == 0) {
      if (int rc = prep_first()) return rc;
      if (n_pages == 1 && !fallback)
        if (int rc = setup_nominate()) return rc;
    } else {
      PROQA_HIP(launch_prep_queries(xq_dev, dtype, nq, idx->ws_nq_pad, idx->xq_pad, idx->tau, idx->run_n, idx->stat_dev,
                                    idx->done, false, nullptr, idx->overflow, st));
    }
    if (idx->exact)
      PROQA_HIP(launch_query_margins(xq_dev, dtype, nq, idx->ws_nq_pad, idx->norm_stats, idx->xq32, idx->margin,
                                     idx->tau, idx->tau_filter, st));
    const PageOut out{D_dev, (long long*)I_dev, idx_offset, k, p * page_size};
    bool boot_overflow = false;
    if (int rc = search_page(idx, qw, n_qtiles, nq, nq_pad, page_k, p > 0, out, st, &fallback, true, &boot_overflow)) return rc;
    if (boot_overflow) {
      // the per-query state is rebuilt from scratch (p == 0 here: later pages never bootstrap)
      PROQA_HIP(launch_prep_queries(xq_dev, dtype, nq, idx->ws_nq_pad, idx->xq_pad, idx->tau, idx->run_n, idx->stat_dev,
                                    nullptr, true, nullptr, idx->overflow, st));
      ++fallback;
      if (int rc = search_page(idx, qw, n_qtiles, nq, nq_pad, page_k, false, out, st, &fallback, false, &boot_overflow))
        return rc;
    }
    if (p + 1 < n_pages)
      PROQA_HIP(launch_advance_page(idx->run_keys, idx->run_n, nq, page_k, idx->bound_keys, idx->ub, idx->done,
                                    idx->exact ? idx->margin : nullptr, idx->exact ? idx->ub_filter : nullptr, st));
  }
  if (n_pages > 1) PROQA_HIP(hipStreamSynchronize(st));  // the last advance_page

  idx->stats.fallback_rounds = fallback;
  idx->stats.candidates += (int64_t)idx->mirror->candidates;
  (void)hipEventElapsedTime(&idx->stats.total_ms, idx->ev[0], idx->ev[1]);
  note_nomination(idx, nq);
  note_leap(idx);
  return PROQA_OK;
}

// host-side completion of a deferred search: wait for the stream, re-scan what overflowed (rewriting D / I)
int finish_pending(proqa_index* idx, int* rewritten) {
  proqa_index::Pending& pe = idx->pending;
  if (rewritten) *rewritten = 0;
  if (!pe.active) return PROQA_OK;
  pe.active = false;
  PROQA_ON_DEVICE(idx->device);
  PROQA_HIP(hipStreamSynchronize(pe.st));
  PagePlan plan;
  plan.slabs.swap(pe.slabs);
  plan.boot = pe.boot;
  const PageOut out{pe.D, pe.I, pe.idx_offset, pe.k, 0};
  int fallback = 0;
  bool boot_overflow = false;
  if (int rc = page_complete(idx, pe.qw, pe.n_qtiles, pe.nq, pe.nq_pad, pe.k, false, out, pe.st, plan, &fallback, &boot_overflow))
    return rc;
  if (boot_overflow) {
    PROQA_HIP(launch_prep_queries(pe.xq_dev, pe.dtype, pe.nq, idx->ws_nq_pad, idx->xq_pad, idx->tau, idx->run_n, idx->stat_dev,
                                  nullptr, true, nullptr, idx->overflow, pe.st));
    ++fallback;
    if (int rc = search_page(idx, pe.qw, pe.n_qtiles, pe.nq, pe.nq_pad, pe.k, false, out, pe.st, &fallback, false, &boot_overflow))
      return rc;
  }
  idx->stats.fallback_rounds = fallback;
  idx->stats.candidates += (int64_t)idx->mirror->candidates;
  (void)hipEventElapsedTime(&idx->stats.total_ms, idx->ev[0], idx->ev[1]);
  note_nomination(idx, pe.nq);
  note_leap(idx);
  if (rewritten) *rewritten = fallback != 0;
  if (!idx->rescue.ids.empty()) {   // (queries a leaping round left short: searched again, their rows rewritten)
    if (int rc = rescue_short_queries(idx)) return rc;
    if (rewritten) *rewritten = 1;
  }
  // the enqueued search found no current int8 copy of the rows and ran on the fp16 rows: build the copy now, behind the
  // host wait this call is anyway, so that the next search scans it
  if (idx->q8_build_due) {
    idx->q8_build_due = false;
    if (nomination_eligible(idx, pe.k))
      if (int rc = ensure_q8(idx, pe.st)) return rc;
    idx->stats.nomination_state = nomination_state_of(idx);
  }
  return PROQA_OK;
}

}  // namespace
}  // namespace proqa

using namespace proqa;

extern "C" {

int proqa_index_search_begin_device(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k, int64_t idx_offset,
                                    float* D_dev, int64_t* I_dev, uint32_t* status_dev, void* stream) {
  if (!idx || (nq > 0 && (!xq_dev || !D_dev || !I_dev))) return fail(PROQA_EINVAL, "index_search_begin_device: NULL argument");
  return search_device(idx, xq_dev, nq, dtype, k, idx_offset, D_dev, I_dev, as_stream(stream), true, status_dev);
}

int proqa_index_search_finish(proqa_index* idx, int* rewritten) {
  if (!idx) return fail(PROQA_EINVAL, "index_search_finish: NULL handle");
  return finish_pending(idx, rewritten);
}

int proqa_index_create(int d, int64_t capacity_rows, proqa_index** out) {
  if (!out) return fail(PROQA_EINVAL, "index_create: out is NULL");
  *out = nullptr;
  if (d != kDim) return fail(PROQA_EINVAL, "index_create: d=%d, only d=128 is supported (ProQA embeds are 128-d)", d);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PROQA_ENOGPU, "index_create: no HIP device visible");
  proqa_index* idx = new (std::nothrow) proqa_index();
  if (!idx) return fail(PROQA_ENOMEM, "index_create: out of host memory");
  PROQA_HIP(hipGetDevice(&idx->device));
  for (auto& e : idx->ev) PROQA_HIP(hipEventCreate(&e));
  PROQA_HIP(hipMalloc((void**)&idx->overflow, kMaxRounds * sizeof(unsigned)));
  PROQA_HIP(hipMalloc((void**)&idx->inexact, 2 * sizeof(unsigned)));
  PROQA_HIP(hipHostMalloc((void**)&idx->mirror, sizeof(SearchMirror), hipHostMallocMapped | hipHostMallocCoherent));
  memset(idx->mirror, 0, sizeof(SearchMirror));
  if (capacity_rows > 0) {
    if (int rc = reserve_rows(idx, capacity_rows)) {
      proqa_index_free(idx);
      return rc;
    }
  }
  *out = idx;
  return PROQA_OK;
}

int proqa_index_free(proqa_index* idx) {
  if (!idx) return PROQA_OK;
  if (idx->pending.active) (void)hipStreamSynchronize(idx->pending.st);   // a begun search nobody finished
  free_workspace(idx);
  if (idx->xb && idx->owns_xb) (void)hipFree(idx->xb);
  if (idx->overflow) (void)hipFree(idx->overflow);
  if (idx->inexact) (void)hipFree(idx->inexact);
  if (idx->xb32) (void)hipFree(idx->xb32);
  if (idx->norm_stats) (void)hipFree(idx->norm_stats);
  if (idx->mirror) (void)hipHostFree(idx->mirror);
  if (idx->stage_dev) (void)hipFree(idx->stage_dev);
  if (idx->boot_scores) (void)hipFree(idx->boot_scores);
  if (idx->xb8) (void)hipFree(idx->xb8);
  if (idx->blk8) (void)hipFree(idx->blk8);
  if (idx->col) (void)hipFree(idx->col);
  if (idx->col_partial) (void)hipFree(idx->col_partial);
  if (idx->qstats) (void)hipFree(idx->qstats);
  if (idx->stage_pinned) (void)hipHostFree(idx->stage_pinned);
  if (idx->io_stream) (void)hipStreamDestroy(idx->io_stream);
  for (auto& e : idx->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : idx->ev_filter)
    if (e) (void)hipEventDestroy(e);
  delete idx;
  return PROQA_OK;
}

int proqa_index_reset(proqa_index* idx) {
  if (!idx) return fail(PROQA_EINVAL, "index_reset: NULL handle");
  if (!idx->owns_xb) {
    idx->xb = nullptr;
    idx->capacity = 0;
    idx->owns_xb = true;
  }
  idx->n = 0;
  idx->rows32 = 0;
  ++idx->rows_epoch;
  idx->exact = false;   // the float32 buffer stays allocated for the next use
  return PROQA_OK;
}

int proqa_index_ntotal(const proqa_index* idx, int64_t* n) {
  if (!idx || !n) return fail(PROQA_EINVAL, "index_ntotal: NULL argument");
  *n = idx->n;
  return PROQA_OK;
}

int proqa_index_configure(proqa_index* idx, int first_slab_rows, int growth) {
  if (!idx) return fail(PROQA_EINVAL, "index_configure: NULL handle");
  if (first_slab_rows < 0 || growth < 0) return fail(PROQA_EINVAL, "index_configure: negative argument");
  if (first_slab_rows) idx->first_slab_rows = first_slab_rows;
  if (growth) idx->growth = growth;
  return PROQA_OK;
}

int proqa_index_configure_bootstrap(proqa_index* idx, int rows) {
  if (!idx) return fail(PROQA_EINVAL, "index_configure_bootstrap: NULL handle");
  if (rows < 0 || rows > kBootstrapMaxRows || rows % 32)
    return fail(PROQA_EINVAL, "index_configure_bootstrap: rows=%d (0 disables; a multiple of 32 up to %d)", rows,
                kBootstrapMaxRows);
  idx->bootstrap_rows = rows;
  idx->bootstrap_auto = false;
  return PROQA_OK;
}

int proqa_index_configure_nomination(proqa_index* idx, int mode) {
  if (!idx) return fail(PROQA_EINVAL, "index_configure_nomination: NULL handle");
  if (mode < 0 || mode > 2) return fail(PROQA_EINVAL, "index_configure_nomination: mode=%d (0 off, 1 automatic, 2 always)", mode);
  idx->nominate_mode = mode;
  return PROQA_OK;
}

int proqa_index_configure_leap(proqa_index* idx, int mode) {
  if (!idx) return fail(PROQA_EINVAL, "index_configure_leap: NULL handle");
  if (mode < 0 || mode > 1) return fail(PROQA_EINVAL, "index_configure_leap: mode=%d (0 off, 1 automatic)", mode);
  idx->leap_mode = mode;
  idx->leap_pause = idx->leap_skip = idx->leap_strikes = 0;
  return PROQA_OK;
}

int proqa_leap_plan(int64_t rows, int64_t bootstrap_rows, int k, int64_t queries, int nominating, int* rounds, int* rank,
                    double* rows_per_round, double* shortfall_probability) {
  if (!rounds || !rank) return fail(PROQA_EINVAL, "leap_plan: NULL output");
  if (rows <= 0 || bootstrap_rows <= 0 || k <= 0 || queries <= 0) return fail(PROQA_EINVAL, "leap_plan: non-positive argument");
  const LeapPlan lp = plan_leap(rows, bootstrap_rows, k, queries > 256 ? 2 : 1, nominating != 0);
  *rounds = lp.rounds;
  *rank = lp.rank;
  if (rows_per_round) *rows_per_round = lp.per_round;
  if (shortfall_probability)
    *shortfall_probability = lp.rounds ? leap_fail_probability(k, lp.rank, std::pow((double)rows / (double)bootstrap_rows, 1.0 / lp.rounds)) : 0.0;
  return PROQA_OK;
}

int proqa_index_prepare(proqa_index* idx, void* stream) {
  if (!idx) return fail(PROQA_EINVAL, "index_prepare: NULL handle");
  if (idx->pending.active) return fail(PROQA_EINVAL, "index_prepare: a search begun on this handle has not been finished");
  PROQA_ON_DEVICE(idx->device);
  hipStream_t st = as_stream(stream);
  idx->q8_build_due = false;
  if (nomination_eligible(idx, 1))
    if (int rc = ensure_q8(idx, st)) return rc;
  PROQA_HIP(hipStreamSynchronize(st));
  return PROQA_OK;
}

int proqa_index_rows_changed(proqa_index* idx) {
  if (!idx) return fail(PROQA_EINVAL, "index_rows_changed: NULL handle");
  if (idx->pending.active) return fail(PROQA_EINVAL, "index_rows_changed: a search begun on this handle has not been finished");
  if (idx->exact) return fail(PROQA_EINVAL, "index_rows_changed: the index keeps float32 copies of its rows (exact-float32 mode); "
                                            "reset it and add the rows again");
  ++idx->rows_epoch;
  return PROQA_OK;
}

int proqa_index_is_exact_f32(const proqa_index* idx, int* enabled) {
  if (!idx || !enabled) return fail(PROQA_EINVAL, "index_is_exact_f32: NULL argument");
  *enabled = idx->exact ? 1 : 0;
  return PROQA_OK;
}

int proqa_index_allow_rounding(proqa_index* idx, int allow) {
  if (!idx) return fail(PROQA_EINVAL, "index_allow_rounding: NULL handle");
  idx->allow_rounding = allow != 0;
  return PROQA_OK;
}

int proqa_index_set_profiling(proqa_index* idx, int enable) {
  if (!idx) return fail(PROQA_EINVAL, "index_set_profiling: NULL handle");
  if (enable && !idx->ev_filter[0]) {
    for (auto& e : idx->ev_filter) PROQA_HIP(hipEventCreate(&e));
  }
  idx->profile = enable != 0;
  return PROQA_OK;
}

int proqa_index_last_stats(const proqa_index* idx, proqa_search_stats* out) {
  if (!idx || !out) return fail(PROQA_EINVAL, "index_last_stats: NULL argument");
  *out = idx->stats;
  return PROQA_OK;
}

int proqa_index_add_device(proqa_index* idx, const void* xb_dev, int64_t n, int dtype, void* stream) {
  if (!idx || (!xb_dev && n > 0) || n < 0) return fail(PROQA_EINVAL, "index_add_device: bad argument");
  if (dtype != PROQA_F16 && dtype != PROQA_F32) return fail(PROQA_EINVAL, "index_add_device: bad dtype %d", dtype);
  if (n == 0) return PROQA_OK;
  PROQA_ON_DEVICE(idx->device);
  if (int rc = reserve_rows(idx, idx->n + n)) return rc;
  hipStream_t st = as_stream(stream);
  char* dst = idx->xb + (size_t)idx->n * kDim * 2;
  if (dtype == PROQA_F16) {
    PROQA_HIP(hipMemcpyAsync(dst, xb_dev, (size_t)n * kDim * 2, hipMemcpyDeviceToDevice, st));
  } else {
    PROQA_HIP(hipMemsetAsync(idx->inexact, 0, 2 * sizeof(unsigned), st));
    PROQA_HIP(launch_convert_f32_to_f16((const float*)xb_dev, dst, n * kDim, idx->inexact, st));
    unsigned bad = 0;
    if (int rc = read_inexact(idx, "index_add_device", st, &bad)) return rc;
    if (bad && !idx->allow_rounding)
      if (int rc = enable_exact(idx, st)) return rc;
  }
  if (int rc = finish_rows_exact(idx, idx->n, n, dtype == PROQA_F32 ? (const float*)xb_dev : nullptr, st)) return rc;
  PROQA_HIP(hipStreamSynchronize(st));
  idx->n += n;
  ++idx->rows_epoch;
  return PROQA_OK;
}

int proqa_index_adopt_device(proqa_index* idx, const void* xb_dev_f16, int64_t n) {
  if (!idx || !xb_dev_f16 || n <= 0) return fail(PROQA_EINVAL, "index_adopt_device: bad argument");
  if (idx->n != 0) return fail(PROQA_EINVAL, "index_adopt_device: index is not empty");
  if (idx->xb && idx->owns_xb) PROQA_HIP(hipFree(idx->xb));
  idx->xb = (char*)const_cast<void*>(xb_dev_f16);
  idx->owns_xb = false;
  idx->n = n;
  idx->capacity = n;
  idx->exact = false;
  ++idx->rows_epoch;
  return PROQA_OK;
}

int proqa_index_add(proqa_index* idx, const void* xb, int64_t n, int dtype) {
  if (!idx || (!xb && n > 0) || n < 0) return fail(PROQA_EINVAL, "index_add: bad argument");
  if (dtype != PROQA_F16 && dtype != PROQA_F32) return fail(PROQA_EINVAL, "index_add: bad dtype %d", dtype);
  if (n == 0) return PROQA_OK;
  PROQA_ON_DEVICE(idx->device);
  if (int rc = reserve_rows(idx, idx->n + n)) return rc;
  const size_t esz = dtype == PROQA_F16 ? 2 : 4;
  // upload in bounded pieces (the source may be an mmap of a multi-GB .npy)
  for (int64_t r0 = 0; r0 < n; r0 += kAddPieceRows) {
    const int64_t m = std::min<int64_t>(kAddPieceRows, n - r0);
    const char* src = (const char*)xb + (size_t)r0 * kDim * esz;
    if (dtype == PROQA_F16) {
      PROQA_HIP(hipMemcpy(idx->xb + (size_t)(idx->n + r0) * kDim * 2, src, (size_t)m * kDim * 2, hipMemcpyHostToDevice));
      if (int rc = finish_rows_exact(idx, idx->n + r0, m, nullptr, nullptr)) return rc;
    } else if (int rc = ingest_f32_piece(idx, idx->n + r0, m, src, /*pinned=*/false, nullptr)) {
      return rc;
    }
  }
  idx->n += n;
  ++idx->rows_epoch;
  return PROQA_OK;
}

// np.load + index.add of the reference in one call, without the host copy in between: reader threads pread() the rows
// into a ring of pinned buffers while the pieces read before are on their way over PCIe (see proqa_hip.h).
int proqa_index_add_npy(proqa_index* idx, const char* path, int64_t row0, int64_t n, int n_readers) {
  if (!idx || !path) return fail(PROQA_EINVAL, "index_add_npy: NULL argument");
  proqa_npy_info info;
  if (int rc = proqa_npy_stat(path, &info)) return rc;
  if (info.cols != kDim) return fail(PROQA_EINVAL, "index_add_npy: %s holds %lld-d rows, the index %d-d", path, (long long)info.cols, kDim);
  if (n < 0) n = info.rows - row0;
  if (row0 < 0 || n < 0 || row0 + n > info.rows)
    return fail(PROQA_EINVAL, "index_add_npy: rows [%lld, %lld) of a file of %lld rows", (long long)row0, (long long)(row0 + n),
                (long long)info.rows);
  if (n == 0) return PROQA_OK;
  PROQA_ON_DEVICE(idx->device);
  if (int rc = reserve_rows(idx, idx->n + n)) return rc;
  if (!idx->io_stream) PROQA_HIP(hipStreamCreateWithFlags(&idx->io_stream, hipStreamNonBlocking));
  const bool f32 = info.dtype == PROQA_F32;
  const size_t row_bytes = (size_t)kDim * (f32 ? 4 : 2);
  const int64_t piece_rows = std::min<int64_t>(kLoaderPieceBytes / (int64_t)row_bytes, n);
  const size_t piece_bytes = (size_t)piece_rows * row_bytes;
  const int64_t n_pieces = ceil_div<int64_t>(n, piece_rows);
  const int n_slots = (int)std::min<int64_t>(kLoaderSlots, n_pieces);
  NpyRing ring;
  ring.fd = open(path, O_RDONLY | O_CLOEXEC);
  if (ring.fd < 0) return fail(PROQA_EIO, "index_add_npy: cannot open %s: %s", path, strerror(errno));
  const off_t base = (off_t)info.data_offset + (off_t)row0 * (off_t)row_bytes;
  (void)posix_fadvise(ring.fd, base, (off_t)n * (off_t)row_bytes, POSIX_FADV_SEQUENTIAL);
  // a file that is not in the page cache: ask the kernel for the whole range now (asynchronous read-ahead at the device's
  // queue depth) -- the four readers' interleaved 8 MiB pieces do not look sequential to the per-descriptor read-ahead
  (void)posix_fadvise(ring.fd, base, (off_t)n * (off_t)row_bytes, POSIX_FADV_WILLNEED);
  PROQA_HIP(hipHostMalloc((void**)&ring.pinned, piece_bytes * n_slots, hipHostMallocDefault));
  for (int s = 0; s < n_slots; ++s) PROQA_HIP(hipEventCreateWithFlags(&ring.uploaded[s], hipEventDisableTiming));
  ring.n_slots = n_slots;
  ring.n_pieces = n_pieces;
  // piece p -> slot p % n_slots; a reader may fill it once piece p - n_slots has left the slot (ring.retired counts the
  // pieces, in order, whose upload has completed)
  auto read_piece = [&](int64_t p) -> bool {
    const int64_t m = std::min<int64_t>(piece_rows, n - p * piece_rows);
    char* dst = ring.pinned + (size_t)(p % n_slots) * piece_bytes;
    size_t want = (size_t)m * row_bytes, got = 0;
    const off_t off = base + (off_t)p * (off_t)piece_bytes;
    while (got < want) {
      const ssize_t r = pread(ring.fd, dst + got, want - got, off + (off_t)got);
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) {
        ring.set_error(r < 0 ? strerror(errno) : "file is shorter than its header says");
        return false;
      }
      got += (size_t)r;
    }
    return true;
  };
  auto reader = [&]() {
    for (;;) {
      int64_t p;
      {
        std::unique_lock<std::mutex> lk(ring.m);
        p = ring.next_piece;
        if (p >= n_pieces || ring.failed) return;
        ++ring.next_piece;
        ring.cv.wait(lk, [&] { return ring.failed || p < ring.retired + n_slots; });
        if (ring.failed) return;
      }
      const bool ok = read_piece(p);
      std::lock_guard<std::mutex> lk(ring.m);
      if (ok) ring.filled[p % n_slots] = p;
      ring.cv.notify_all();
      if (!ok) return;
    }
  };
  if (n_readers <= 0) n_readers = 4;
  n_readers = (int)std::min<int64_t>(n_readers, std::min<int64_t>(n_pieces, kLoaderSlots));
  try {
    for (int t = 0; t < n_readers; ++t) ring.threads.emplace_back(reader);
  } catch (...) {
    // (thread limit of a cgroup: the threads that did start keep reading; with none, this thread reads in line)
  }
  const bool inline_reads = ring.threads.empty();
  hipStream_t st = idx->io_stream;
  ring.stream = st;
  const int in_flight = std::max(1, n_slots / 2);   // uploads on the stream before the host waits for the oldest
  for (int64_t p = 0; p < n_pieces; ++p) {
    const int slot = (int)(p % n_slots);
    if (inline_reads) {
      if (!read_piece(p)) break;
    } else {
      std::unique_lock<std::mutex> lk(ring.m);
      ring.cv.wait(lk, [&] { return ring.failed || ring.filled[slot] == p; });
      if (ring.failed) break;
    }
    const int64_t m = std::min<int64_t>(piece_rows, n - p * piece_rows);
    const int64_t row = idx->n + p * piece_rows;
    const char* src = ring.pinned + (size_t)slot * piece_bytes;
    int rc = PROQA_OK;
    if (!f32) {
      if (hipError_t e = hipMemcpyAsync(idx->xb + (size_t)row * kDim * 2, src, (size_t)m * row_bytes, hipMemcpyHostToDevice, st))
        rc = hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
      else
        rc = finish_rows_exact(idx, row, m, nullptr, st);
    } else {
      rc = ingest_f32_piece(idx, row, m, src, /*pinned=*/true, st);
    }
    if (rc == PROQA_OK)
      if (hipError_t e = hipEventRecord(ring.uploaded[slot], st)) rc = hip_fail(e, "hipEventRecord", __FILE__, __LINE__);
    if (rc == PROQA_OK && p + 1 >= in_flight) {
      const int64_t done = p + 1 - in_flight;      // the oldest upload in flight: its slot goes back to the readers
      if (hipError_t e = hipEventSynchronize(ring.uploaded[done % n_slots])) rc = hip_fail(e, "hipEventSynchronize", __FILE__, __LINE__);
      std::lock_guard<std::mutex> lk(ring.m);
      ring.retired = done + 1;
      ring.cv.notify_all();
    }
    if (rc != PROQA_OK) {
      ring.set_error(nullptr);
      return rc;               // (~NpyRing joins the readers and releases the ring)
    }
  }
  bool read_failed;
  {
    std::lock_guard<std::mutex> lk(ring.m);
    read_failed = ring.failed;
  }
  if (read_failed) {
    (void)hipStreamSynchronize(st);
    return fail(PROQA_EIO, "index_add_npy: reading %s failed: %s", path, ring.why);
  }
  PROQA_HIP(hipStreamSynchronize(st));
  idx->n += n;
  ++idx->rows_epoch;
  return PROQA_OK;
}

int proqa_index_search_device(proqa_index* idx, const void* xq_dev, int64_t nq, int dtype, int k,
                              int64_t idx_offset, float* D_dev, int64_t* I_dev, void* stream) {
  if (!idx || (nq > 0 && (!xq_dev || !D_dev || !I_dev))) return fail(PROQA_EINVAL, "index_search_device: NULL argument");
  return search_device(idx, xq_dev, nq, dtype, k, idx_offset, D_dev, I_dev, as_stream(stream));
}

int proqa_index_search(proqa_index* idx, const void* xq, int64_t nq, int dtype, int k, float* D, int64_t* I) {
  if (!idx || (nq > 0 && (!xq || !D || !I))) return fail(PROQA_EINVAL, "index_search: NULL argument");
  if (dtype != PROQA_F16 && dtype != PROQA_F32) return fail(PROQA_EINVAL, "index_search: bad dtype %d", dtype);
  if (nq == 0) return PROQA_OK;
  if (k <= 0) return fail(PROQA_EINVAL, "index_search: k=%d", k);
  PROQA_ON_DEVICE(idx->device);
  // host buffers cross through one pinned staging area on the index's own stream: queries up, the search, results
  // down, ONE synchronisation (pageable hipMemcpy calls on the null stream would each stage and synchronise)
  const size_t esz = dtype == PROQA_F16 ? 2 : 4;
  const size_t q_bytes = round_up<size_t>((size_t)nq * kDim * esz, 256);
  const size_t d_bytes = round_up<size_t>((size_t)nq * k * sizeof(float), 256);
  const size_t i_bytes = (size_t)nq * k * sizeof(int64_t);
  const size_t total = q_bytes + d_bytes + i_bytes;
  if (int rc = ensure_stage(idx, total)) return rc;
  if (total > idx->stage_pinned_bytes) {
    if (idx->stage_pinned) PROQA_HIP(hipHostFree(idx->stage_pinned));
    idx->stage_pinned = nullptr;
    idx->stage_pinned_bytes = 0;
    PROQA_HIP(hipHostMalloc(&idx->stage_pinned, total, hipHostMallocDefault));
    idx->stage_pinned_bytes = total;
  }
  if (!idx->io_stream) PROQA_HIP(hipStreamCreateWithFlags(&idx->io_stream, hipStreamNonBlocking));
  hipStream_t st = idx->io_stream;
  char* dev = (char*)idx->stage_dev;
  char* host = (char*)idx->stage_pinned;
  memcpy(host, xq, (size_t)nq * kDim * esz);
  PROQA_HIP(hipMemcpyAsync(dev, host, (size_t)nq * kDim * esz, hipMemcpyHostToDevice, st));
  float* D_dev = (float*)(dev + q_bytes);
  int64_t* I_dev = (int64_t*)(dev + q_bytes + d_bytes);
  if (int rc = search_device(idx, dev, nq, dtype, k, 0, D_dev, I_dev, st)) return rc;
  PROQA_HIP(hipMemcpyAsync(host + q_bytes, dev + q_bytes, d_bytes + i_bytes, hipMemcpyDeviceToHost, st));
  PROQA_HIP(hipStreamSynchronize(st));
  memcpy(D, host + q_bytes, (size_t)nq * k * sizeof(float));
  memcpy(I, host + q_bytes + d_bytes, i_bytes);
  return PROQA_OK;
}

int proqa_index_reconstruct_batch_device(proqa_index* idx, const int64_t* ids_dev, int64_t n, int64_t idx_offset, void* out_dev,
                                         int out_dtype, void* stream) {
  if (!idx) return fail(PROQA_EINVAL, "index_reconstruct_batch: idx is NULL");
  if (n < 0 || (n > 0 && (!ids_dev || !out_dev))) return fail(PROQA_EINVAL, "index_reconstruct_batch: n=%lld or NULL argument", (long long)n);
  if (out_dtype != PROQA_F16 && out_dtype != PROQA_F32) return fail(PROQA_EINVAL, "index_reconstruct_batch: bad dtype %d", out_dtype);
  PROQA_ON_DEVICE(idx->device);
  PROQA_HIP(launch_gather_index_rows(idx->xb, idx->exact ? idx->xb32 : nullptr, idx->n, (const long long*)ids_dev, n, idx_offset,
                                     out_dev, out_dtype == PROQA_F32, (hipStream_t)stream));
  return PROQA_OK;
}

int proqa_topk_merge_strided_device(const float* D_parts_dev, const int64_t* I_parts_dev, int n_parts, int64_t nq, int k,
                                    int64_t stride_d, int64_t stride_i, float* D_dev, int64_t* I_dev, void* stream) {
  if (!D_parts_dev || !I_parts_dev || !D_dev || !I_dev || n_parts <= 0 || nq < 0 || k <= 0)
    return fail(PROQA_EINVAL, "topk_merge_device: bad argument");
  if ((long long)n_parts * k >= (1ll << 27))
    return fail(PROQA_EINVAL, "topk_merge_device: n_parts*k=%lld is too large", (long long)n_parts * k);
  if (stride_d < nq * k || stride_i < nq * k)
    return fail(PROQA_EINVAL, "topk_merge_device: part strides %lld / %lld are shorter than a part (%lld)", (long long)stride_d,
                (long long)stride_i, (long long)(nq * k));
  // (lists of unknown provenance: always the sorting kernels, never the rank merge that trusts the order of its parts)
  PROQA_HIP(launch_merge_lists(D_parts_dev, (const long long*)I_parts_dev, n_parts, nq, k, (long long)stride_d, (long long)stride_i,
                               D_dev, (long long*)I_dev, as_stream(stream), nullptr, 0, nullptr, /*parts_sorted=*/false));
  return PROQA_OK;
}

int proqa_topk_merge_device(const float* D_parts_dev, const int64_t* I_parts_dev, int n_parts, int64_t nq,
                            int k, float* D_dev, int64_t* I_dev, void* stream) {
  return proqa_topk_merge_strided_device(D_parts_dev, I_parts_dev, n_parts, nq, k, nq * k, nq * k, D_dev, I_dev, stream);
}

}  // extern "C"

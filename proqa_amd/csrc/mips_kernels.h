// Launch interface between mips_index.cpp (host orchestration) and mips_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/proqa_hip.h"

namespace proqa {

constexpr int kDim = PROQA_EMBED_DIM;  // 128
constexpr int kFilterWaves = 8;
constexpr int kFilterThreads = kFilterWaves * 64;
// queries of one filter workgroup: 8 waves x qw blocks of 32 (qw = 1, 2), or 4 waves x 4 blocks (qw = 4)
__host__ __device__ constexpr unsigned filter_tile_queries(int qw) { return (qw == 4 ? 4u : 8u) * (unsigned)qw * 32u; }
constexpr int kStageRows = 128;  // corpus rows per LDS stage (32 KiB of fp16 rows)
constexpr int kMergeThreads = 256;
// merge_lists sorts up to this many gathered keys per query in LDS (128 KiB); above it the merge
// runs as a segmented radix sort in HBM
constexpr int kMaxMergeListKeys = 16384;
constexpr int kMaxSortKeys = 2048;  // running list + candidates of one query per merge (LDS), k <= kPageK
constexpr int kPageK = kMaxSortKeys / 2;  // k up to here is one page at the small merge capacity
// k > kPageK (retrieval/trec_process.py:76 asks for 10000, qa/online_sampler.py:113 for 5000): pages of kBigPageK
// results on a merge that holds kBigSortKeys keys (64 KiB of LDS), with deeper lane lists and more corpus chunks
constexpr int kBigSortKeys = 8192;
constexpr int kMidSortKeys = 4096;  // the sample round of the one-pass search (~2 k candidates per query): four workgroups per CU
constexpr int kBigPageK = kBigSortKeys / 2;
// One-pass search of a large k (see search_one_pass in mips_index.cpp): ONE filter launch over the whole shard against
// thresholds estimated from a sample, and ONE merge per query that holds up to kOnePassSortKeys keys (128 KiB of LDS)
constexpr int kOnePassSortKeys = 16384;
constexpr int kOnePassMergeThreads = 1024;  // of that merge (the others run kMergeThreads)
constexpr int kOnePassLaneCap = 24;  // records per lane list of that launch (the chunk count aims at ~8 per list)
constexpr int kLaneCap = 8;         // records a lane can log per (chunk, query) before spilling
constexpr int kBigLaneCap = 32;     // the same for big pages (thousands of candidates per query and round)
constexpr int kBigMinChunks = 256;  // corpus chunks of a big-page launch (>= 512 lane lists per query)
__host__ __device__ constexpr int sort_capacity(int page_k) { return page_k <= kPageK ? kMaxSortKeys : kBigSortKeys; }
__host__ __device__ constexpr int lane_capacity(int page_k) { return page_k <= kPageK ? kLaneCap : kBigLaneCap; }
constexpr int kBootstrapMaxRows = 8192;  // rows the bootstrap can cover (32 keys per thread)
constexpr int kBootstrapMaxK = 256;       // its bound is the k-th of 256 thread maxima
constexpr int kSpillCap = 256;      // shared spill records per (chunk, wave)

// One lane's 16-score accumulator column that beat its query's threshold.  The filter kernel
// logs whole columns (one 80-byte store burst, no per-score work in the MFMA loop); the merge
// kernel picks the individual scores that pass.
struct __attribute__((aligned(16))) WaveRecord {
  unsigned q;          // query index (padded numbering)
  unsigned row0;       // shard-local corpus row of score[0]
  int rows_left;       // score[r] is a real row iff (r&3) + 8*(r>>2) < rows_left
  float tau;           // the threshold the column was tested against
  float score[16];     // accumulator registers: row offset of r is (r&3) + 8*(r>>2)
};
static_assert(sizeof(WaveRecord) == 80, "record layout");

// Candidate storage of one filter launch (all written without atomics):
//   lane_log [chunk][q][half][kLaneCap]  private list of the lane that owns (q, half) in `chunk`
//   lane_cnt [q][chunk][half]            its length (query-major: one contiguous run per query for the merge)
//   spill_log[chunk][wave slot][kSpillCap], spill_cnt[wave slot][chunk]
//                                         shared by the 64 lanes of a wave once a private list is full
// wave slot = (query tile, wave) = the 32*QW consecutive queries one wave owns.
struct CandidateStore {
  WaveRecord* lane_log;
  unsigned* lane_cnt;
  WaveRecord* spill_log;
  unsigned* spill_cnt;
  unsigned nq_pad;
  unsigned n_qtiles;
  unsigned lane_cap;     // records per lane list of this search (kLaneCap or kBigLaneCap)
  unsigned n_chunks;     // corpus chunks of this launch, rounded up to 8: stride of the query-major / slot-major counters
};

constexpr int kKeysPerRecord = 10;  // compact lists: 8-byte keys in the room of one WaveRecord
static_assert(kKeysPerRecord * 8 == 80, "compact list addressing");
constexpr int kCompactLaneCap = 7;  // records' worth per lane list of a compact launch: room for 70 keys (the chunk count aims at ~24)
constexpr unsigned kCompactKeys = 64;  // keys a compact list holds: a power of two, so that the append position is n & 63 without a capacity branch
static_assert(kCompactKeys <= kCompactLaneCap * kKeysPerRecord, "compact list size");

struct FilterArgs {
  const void* xq;        // fp16 [nq_pad,128], zero rows beyond nq
  const char* xb;        // fp16 corpus rows of this shard
  long long slab_row0;   // rows [slab_row0, slab_row1) are scanned by this launch
  long long slab_row1;
  int rows_per_chunk;    // multiple of kStageRows; one workgroup per (chunk, query tile)
  const float* tau;      // running k-th best score per query (-inf until k rows were seen)
  const float* ub;       // paged (k > kPageK) searches: scores above ub[q] were reported by an earlier page
  CandidateStore store;
  unsigned* overflow;    // set to 1 if a record had to be dropped in this launch
  unsigned flags;        // developer experiments (PROQA_FILTER_FLAGS), 0 in production
  int compact;           // lane lists hold the passing scores as 8-byte keys (mips_filter_f16<COMPACT>), not column records
};

struct MergeArgs {
  CandidateStore store;
  unsigned n_chunks;             // chunks of the filter launch that scanned rows
  unsigned qw;                   // query blocks per wave of that launch
  unsigned long long* run_keys;  // [nq_pad, k] sorted descending
  unsigned* run_n;               // valid entries per query
  float* tau;
  int k;
  int sort_cap;                  // kMaxSortKeys, kBigSortKeys or kOnePassSortKeys: keys one merge holds (selects the kernel instantiation)
  int inclusive;                 // overflow-safe rounds: >= threshold, duplicates removed
  const unsigned long long* bound_keys;  // paged searches: only keys strictly below bound_keys[q] count (or NULL)
  unsigned long long* stat_candidates;  // [nq_pad] candidates merged per query (statistics)
  unsigned* overflow;
  // exact-float32 mode (all NULL otherwise): the filter ran on the fp16 roundings against
  // tau_filter = tau - margin; every logged score that passed is re-scored from the float32 rows
  // (double accumulation, rounded once) and compared with the exact threshold `tau`
  const float* xq32;             // [nq_pad,128] float32 queries
  const float* xb32;             // float32 corpus rows of this shard
  const float* margin;           // [nq_pad] bound on |float32 score - fp16 score| for any row
  float* tau_filter;             // [nq_pad] threshold of the next filter launch
  unsigned long long* dbg;       // developer build (PROQA_MERGE_STAMPS): [nq_pad, 8] s_memtime stamps of the phases, or NULL
  int compact;                   // the launch logged compact lists (FilterArgs::compact)
  // int8 nomination rounds (all NULL otherwise): the records hold int32 scores and the integer threshold they were tested
  // against; every score above it names a row that is re-scored from xb16 against xq16 on the filter's MFMA sequence
  int nom_keys;                  // keys the merge of this nominating round may hold: 1024 (rounds that nominate few rows) or 2048
  const void* xq16;              // fp16 [nq_pad,128] padded queries
  const char* xb16;              // fp16 corpus rows of this shard
  unsigned long long* stat_nominated;  // [nq_pad] rows re-scored per query (statistics)
  // leaping rounds (see topk_merge): rank (1-based, < k) of the running list whose score becomes the next round's threshold, 0 =
  // the k-th best; leap_check: this round's scan tested against such a rank -- verify that k keys reach it (else overflow bit 3)
  int next_rank;
  int leap_check;
  unsigned* short_rounds;        // [nq_pad] bit round_bit set: the query ended that leaping round with fewer than k rows above its threshold
  int round_bit;
};

// ---- int8 nomination scan (the k <= kPageK rounds of an fp16 index; see "int8 nomination" in mips_kernels.hip) ----------
// An int8 copy of the centred, per-dimension- and per-32-row-block-scaled rows is scanned on v_mfma_i32_32x32x32_i8 (twice the
// fp16 rate, exact i32 sums); a row is NOMINATED when its integer score exceeds the query's running threshold lowered by a
// rigorous bound on the quantisation error; the merge re-scores the nominated rows from the fp16 rows with the filter's own
// MFMA sequence (bit-identical scores) and keeps those that beat the exact threshold.
constexpr int kStageBytesI8 = kStageRows * kDim;   // 16 KiB: a 128-row stage of int8 rows
struct NominateParams {     // per query, written by prep_queries_i8 (threshold of block b: A G_b - B - E X_b, A = (tau - off) / s_q)
  float off;                // q . mean (the centring constant of this query's scores) + the slack in score units
  float inv_unit;           // 1 / s_q
  float margin_r;           // B = ||w / s_q|| R: the rows' rounding residuals, in integer units
  float err_norm;           // E = ||e||: the query's own rounding residual (times the block's max ||xi||)
};
static_assert(sizeof(NominateParams) == 16, "one 16-byte load per lane");
struct QuantStats {         // device words maintained by the quantisation passes (floats as bits: atomicMax on non-negatives)
  unsigned max_resid;       // R  = max over rows of ||127 (x - mean) / c - xi||
  unsigned max_inorm;       // max over rows of ||xi|| (per block: blk[b].y)
  unsigned max_xnorm;       // Xf = max over rows of ||x|| (fp16 rows)
  unsigned nonfinite;       // rows or statistics that are not finite: the scan is not used
};
struct FilterArgsI8 {
  const signed char* xq8;   // int8 [nq_pad,128] queries (prep_queries_i8), zero rows beyond nq
  const signed char* xb8;   // int8 corpus rows of this shard
  long long slab_row0;
  long long slab_row1;
  int rows_per_chunk;       // multiple of kStageRows
  const float* tau;         // exact running k-th best score per query (+inf: padding / exhausted, -inf: fewer than k rows yet)
  const NominateParams* qp; // [nq_pad]
  const float2* blk;        // per 32-row block of the shard: {G_b = 127 / f_b, X_b = max ||xi|| of its rows}; readable for 8
                            // blocks past the shard's last one (the scan fetches whole pairs of stages)
  CandidateStore store;     // lane lists of 8-byte records {first row of the column, 16 nominee bits} in the column records' slots
  unsigned* overflow;
  unsigned flags;           // developer cut experiments (PROQA_FILTER_FLAGS; wrong results), 0 in production
  unsigned q_blocks;        // row-split launches (few queries): 32-query blocks that hold queries -- 1, 2 or 4; else 0.  The 8 /
                            // q_blocks waves that share a query block take every (8 / q_blocks)-th 32-row unit each and append
                            // to the block's lists through workgroup-shared counters
};
hipError_t launch_filter_i8(const FilterArgsI8& a, int qw, unsigned grid, hipStream_t st, bool deep = false);
// column statistics of fp16 rows [0, n): partial[g][0..127] sums, [g][128..255] minima, [g][256..383] maxima per workgroup g
// (deterministic two-level reduction), then mean / scale per dimension: col[0..127] mean, col[128..255] 1 / c, col[256..383] c
constexpr int kColStatGroups = 1024;
hipError_t launch_column_stats(const void* xb16, long long n, float* partial, float* col, QuantStats* stats, hipStream_t st);
// xi = clamp(rint(127 (x - mean) / (c f_b))) with the block scales f_b, blk[(n + 31) / 32], and the maxima of QuantStats
hipError_t launch_quantise_rows_i8(const void* xb16, long long n, const float* col, signed char* xb8, float2* blk, QuantStats* stats,
                                   hipStream_t st);
// int8 queries + NominateParams of every padded query (fp16 padded queries in, as the filter reads them)
// (also zeroes stat_nom[0, nq_pad))
hipError_t launch_prep_queries_i8(const void* xq_pad16, long long nq_pad, const float* col, const QuantStats* stats,
                                  signed char* xq8, NominateParams* qp, unsigned long long* stat_nom, hipStream_t st);

hipError_t launch_filter(const FilterArgs& a, int qw, bool inclusive, unsigned grid, hipStream_t st);
#ifdef PROQA_FILTER_STAMPS
void read_filter_stamps(unsigned long long* out5);   // developer build: sums of the s_memtime stamps, then reset
#endif
// margin/ub_filter: exact-float32 mode (NULL otherwise): ub_filter[q] = ub[q] + margin[q]
hipError_t launch_advance_page(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int page_k,
                               unsigned long long* bound_keys, float* ub, unsigned char* done, const float* margin,
                               float* ub_filter, hipStream_t st);
hipError_t launch_merge(const MergeArgs& a, unsigned nq_pad, hipStream_t st);
// out[r] = row ids[r] - idx_offset of the index (zero row when outside [0, n_index)); fp16 rows, or float32 (xb32 when the
// index keeps float32 copies, else upcasts of the fp16 rows)
hipError_t launch_gather_index_rows(const void* xb16, const float* xb32, long long n_index, const long long* ids, long long n,
                                    long long idx_offset, void* out, bool out_f32, hipStream_t st);
// *flag = 1 if any of the nq lists holds fewer than `want` keys
hipError_t launch_flag_short_lists(const unsigned* run_n, long long nq, unsigned want, unsigned* flag, hipStream_t st);
// exact top-k of rows [0, n_rows) (n_rows <= kBootstrapMaxRows, k <= kBootstrapMaxK) for every query: run_keys / run_n /
// tau as the geometric rounds would leave them after those rows.  scores: [nq_pad, round_up(n_rows, 32)] floats.
// run_stride (0 = k): keys per query in run_keys when the list that follows is longer than this k (search_one_pass)
hipError_t launch_bootstrap(const char* xb, const void* xq_pad, int n_rows, unsigned nq_pad, int k, float* scores,
                            unsigned long long* run_keys, unsigned* run_n, float* tau, unsigned long long* stat,
                            unsigned* overflow, hipStream_t st, int run_stride = 0, int tau_rank = 0);
// overflow (optional): the kOverflowWords round words of the page that follows are zeroed
hipError_t launch_prep_queries(const void* xq, int dtype, long long nq, long long nq_pad, void* xq_pad,
                               float* tau, unsigned* run_n, unsigned long long* stat, const unsigned char* done,
                               bool reset_stat, unsigned* inexact, unsigned* overflow, hipStream_t st,
                               unsigned* short_rounds = nullptr);
// One-pass search of a large k for a FEW queries (mips_index.cpp one_pass_big_launch_i8): the launch over the shard scans the
// int8 copy (half the bytes of an HBM-bound pass) and logs {row0, nominee bits} records; this kernel re-scores the nominated
// rows of `lists_per_group` lane lists per workgroup from the fp16 rows (the fp16 scan's MFMA sequence: its bits) and writes
// the keys that beat tau as ONE compact list per group -- the lists the compact merge (MergeArgs::compact) sorts.
struct RescoreArgs {
  CandidateStore in;          // records of the int8 launch (8 bytes each, in.lane_cap per list)
  unsigned in_lists;          // 2 x its chunks
  unsigned lists_per_group;   // <= 64
  CandidateStore out;         // group g -> list (chunk g >> 1, half g & 1): kCompactKeys keys each, lengths in out.lane_cnt
  const void* xq16;           // fp16 [nq_pad,128]
  const char* xb16;           // fp16 rows of the shard
  const float* tau;
  unsigned* overflow;         // a group nominated more rows than it holds, or more keys passed than a compact list holds
  unsigned long long* stat_nominated;
};
hipError_t launch_rescore_nominated_lists(const RescoreArgs& a, unsigned groups, unsigned nq, hipStream_t st);
// rows ids[0..n) of the padded fp16 queries -> out [n,128]; result rows [n,k] -> rows ids[i] of D / I (row stride out_stride)
hipError_t launch_gather_query_rows(const void* xq_pad, const int* ids, int n, void* out, hipStream_t st);
hipError_t launch_scatter_result_rows(const float* D_src, const long long* I_src, const int* ids, int n, int k, float* D, long long* I,
                                      int out_stride, hipStream_t st);
// What a search's one host synchronisation reads, written by the finalize kernel straight into pinned host memory (no
// copy command on the stream): the overflow words of the rounds and the candidates summed over the queries.
constexpr int kOverflowWords = 96;   // = kMaxRounds of mips_index.cpp
struct SearchMirror {
  unsigned overflow[kOverflowWords];
  unsigned long long candidates;
  unsigned long long nominated;   // rows the int8 rounds re-scored, summed over the queries (0 for an fp16 search)
};
// writes page results: D/I[q * out_stride + out_offset + j], j < page_k.  overflow: the kOverflowWords round words (device);
// status (optional device word) = 1 if any is set, else 0; mirror (optional, pinned host memory) receives the words and the
// sum of stat[0, nq)
hipError_t launch_finalize(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int page_k,
                           long long idx_offset, float* D, long long* I, int out_stride, int out_offset,
                           const unsigned* overflow, unsigned* status, SearchMirror* mirror, const unsigned long long* stat,
                           hipStream_t st, const unsigned long long* stat_nominated = nullptr);
// part p's [nq, k] scores / ids start stride_d / stride_i ELEMENTS after part p-1's (nq*k for dense [n_parts, nq, k] arrays);
// status_host (optional, pinned host memory, n_parts words) receives status_src[p * status_stride] of every part
// (also when nq == 0).  parts_sorted: every part is a list as a search reports it (scores descending, ties by ascending id,
// no NaN, the I = -1 slots at its tail) -- small merges then rank by binary search instead of sorting; lists of unknown
// order (the public merge entry points) must pass false.
hipError_t launch_merge_lists(const float* D_parts, const long long* I_parts, int n_parts, long long nq,
                              int k, long long stride_d, long long stride_i, float* D, long long* I, hipStream_t st,
                              const unsigned* status_src = nullptr, long long status_stride = 0, unsigned* status_host = nullptr,
                              bool parts_sorted = true);
hipError_t launch_convert_f32_to_f16(const float* src, void* dst, long long n, unsigned* inexact, hipStream_t st);
// exact-float32 mode helpers
hipError_t launch_upconvert_f16_to_f32(const void* src, float* dst, long long n, hipStream_t st);
// norm_stats[0] = max over rows of ||x32 - fp16(x32)||, norm_stats[1] = max ||fp16(x32)|| (float bits, atomicMax)
hipError_t launch_row_norm_stats(const float* xb32, const void* xb16, long long n_rows, unsigned* norm_stats,
                                 hipStream_t st);
// float32 copies of the queries, the per-query error margin, and tau_filter = tau (still +-inf here)
hipError_t launch_query_margins(const void* xq, int dtype, long long nq, long long nq_pad, const unsigned* norm_stats,
                                float* xq32, float* margin, const float* tau, float* tau_filter, hipStream_t st);

}  // namespace proqa

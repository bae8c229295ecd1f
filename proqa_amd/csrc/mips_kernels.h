// Launch interface between mips_index.cpp (host orchestration) and mips_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/proqa_hip.h"

namespace proqa {

constexpr int kDim = PROQA_EMBED_DIM;  // 128
constexpr int kFilterWaves = 8;
constexpr int kFilterThreads = kFilterWaves * 64;
constexpr int kStageRows = 128;  // corpus rows per LDS stage (32 KiB of fp16 rows)
constexpr int kMergeThreads = 256;
constexpr int kMaxSortKeys = 4096;  // k + candidate capacity must fit one LDS sort

// one lane's 16-score accumulator column that beat its query's threshold (see mips_kernels.hip)
struct __attribute__((aligned(16))) WaveRecord {
  unsigned q;          // query index (padded numbering)
  unsigned row0;       // shard-local corpus row of score[0]
  int rows_left;       // score[r] is a real row iff (r&3) + 8*(r>>2) < rows_left
  float tau;           // the threshold the column was tested against
  float score[16];     // accumulator registers: row offset of r is (r&3) + 8*(r>>2)
};
static_assert(sizeof(WaveRecord) == 80, "record layout");

struct FilterArgs {
  const void* xq;        // fp16 [nq_pad,128], zero rows beyond nq
  const char* xb;        // fp16 corpus rows of this shard
  long long slab_row0;   // rows [slab_row0, slab_row1) are scanned by this launch
  long long slab_row1;
  int rows_per_chunk;    // multiple of kStageRows; one workgroup per (chunk, query tile)
  unsigned n_qtiles;
  const float* tau;      // running k-th best score per query (-inf until k rows were seen)
  unsigned* cand_cnt;    // per-query append counter
  uint2* cand;           // [nq_pad, cap] (score bits, shard-local row)
  unsigned cap;
  unsigned* overflow;    // set to 1 if any append was dropped in this launch
  WaveRecord* wave_log;         // [grid * 8 waves, wave_log_cap] private candidate records
  unsigned* wave_log_cnt;       // records written by each wave of this launch
  unsigned wave_log_cap;
};
constexpr int kWaveLogCap = 816;   // records per wave (~64 KiB): > 512 = one dense stage (8 tiles x 64 lanes)

struct MergeArgs {
  uint2* cand;
  unsigned* cand_cnt;
  unsigned cap;
  unsigned long long* run_keys;  // [nq_pad, k] sorted descending
  unsigned* run_n;               // valid entries per query
  float* tau;
  int k;
  int dedupe;
  unsigned long long* stat_candidates;  // [nq_pad] candidates merged per query (statistics)
};

hipError_t launch_filter(const FilterArgs& a, int qw, bool inclusive, unsigned grid, hipStream_t st);
hipError_t launch_merge(const MergeArgs& a, unsigned nq_pad, hipStream_t st);
hipError_t launch_prep_queries(const void* xq, int dtype, long long nq, long long nq_pad, void* xq_pad,
                               float* tau, unsigned* cand_cnt, unsigned* run_n, unsigned long long* stat,
                               hipStream_t st);
hipError_t launch_finalize(const unsigned long long* run_keys, const unsigned* run_n, long long nq, int k,
                           long long idx_offset, float* D, long long* I, hipStream_t st);
hipError_t launch_merge_lists(const float* D_parts, const long long* I_parts, int n_parts, long long nq,
                              int k, float* D, long long* I, hipStream_t st);
hipError_t launch_convert_f32_to_f16(const float* src, void* dst, long long n, hipStream_t st);

}  // namespace proqa

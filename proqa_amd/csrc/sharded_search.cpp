// Row-sharded exact search behind the C ABI (proqa_comm_*, proqa_sharded_search_device in proqa_hip.h).
//
// one process (or thread) per GPU holds rows [lo, hi) of the corpus and all queries; a
// search is the local exact top-k with GLOBAL row ids, ONE RCCL all-gather of the per-rank [nq, k] (id, score)
// lists over xGMI, and the same merge kernel the single-GPU path uses -- bit-identical to searching the whole
// corpus on one GPU.  The reference's only NCCL touch point is /root/reference/retrieval/get_embed.py:44-52
// (process-group init); its search is single-process faiss (eval_retrieval.py:102-104).
//
// RCCL is bound at run time (dlopen/dlsym): the library keeps loading on a box without it, and inside a PyTorch
// process the copy PyTorch already loaded is the one that is used (one RCCL per process).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "common.h"
#include "mips_kernels.h"

namespace proqa {
namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = {0};
};

const RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already in the process first (PyTorch's), then the system one
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      snprintf(api.why, sizeof api.why, "librccl.so could not be loaded: %s", dlerror());
      return;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
      snprintf(api.why, sizeof api.why, "librccl.so lacks an expected ncclXxx symbol");
      return;
    }
    api.ok = true;
  });
  return api;
}

int rccl_fail(ncclResult_t r, const char* what) {
  return fail(PROQA_EHIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
}

constexpr uint32_t kPoison = 0xFFFFFFFFu;   // status word of a rank whose local search failed

#define PROQA_RCCL(call)                                      \
  do {                                                        \
    ncclResult_t _r = (call);                                 \
    if (_r != ncclSuccess) return rccl_fail(_r, #call);       \
  } while (0)

}  // namespace
}  // namespace proqa

struct proqa_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  // exchange buffers (grown on demand): send = this rank's [ids | scores] block, recv = world blocks
  char* send = nullptr;
  char* recv = nullptr;
  size_t block_bytes = 0;
  uint32_t* status_host = nullptr;   // pinned: the status word of every rank's block
};

using namespace proqa;

extern "C" {

int proqa_comm_get_unique_id(void* id_out) {
  if (!id_out) return fail(PROQA_EINVAL, "comm_get_unique_id: NULL argument");
  const RcclApi& api = rccl();
  if (!api.ok) return fail(PROQA_ENOGPU, "%s", api.why);
  static_assert(sizeof(ncclUniqueId) == PROQA_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  PROQA_RCCL(api.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return PROQA_OK;
}

int proqa_comm_create(const void* id, int world_size, int rank, proqa_comm** out) {
  if (!out) return fail(PROQA_EINVAL, "comm_create: out is NULL");
  *out = nullptr;
  if (!id || world_size <= 0 || rank < 0 || rank >= world_size)
    return fail(PROQA_EINVAL, "comm_create: world_size=%d rank=%d", world_size, rank);
  const RcclApi& api = rccl();
  if (!api.ok) return fail(PROQA_ENOGPU, "%s", api.why);
  proqa_comm* c = new (std::nothrow) proqa_comm();
  if (!c) return fail(PROQA_ENOMEM, "comm_create: out of host memory");
  c->world = world_size;
  c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return fail(PROQA_ENOGPU, "comm_create: no current HIP device");
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = api.CommInitRank(&c->comm, world_size, uid, rank);   // collective over the ranks
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail(r, "ncclCommInitRank");
  }
  *out = c;
  return PROQA_OK;
}

int proqa_comm_free(proqa_comm* c) {
  if (!c) return PROQA_OK;
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->status_host) (void)hipHostFree(c->status_host);
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  delete c;
  return PROQA_OK;
}

int proqa_comm_info(const proqa_comm* c, int* world_size, int* rank) {
  if (!c || !world_size || !rank) return fail(PROQA_EINVAL, "comm_info: NULL argument");
  *world_size = c->world;
  *rank = c->rank;
  return PROQA_OK;
}

int proqa_sharded_block_layout(int64_t nq, int k, size_t* ids_bytes, size_t* scores_bytes, size_t* block_bytes) {
  if (nq < 0 || k <= 0 || !ids_bytes || !scores_bytes || !block_bytes) return fail(PROQA_EINVAL, "sharded_block_layout: bad argument");
  *ids_bytes = round_up<size_t>((size_t)nq * k * sizeof(int64_t), 16);
  *scores_bytes = round_up<size_t>((size_t)nq * k * sizeof(float), 16);
  *block_bytes = *ids_bytes + *scores_bytes + 16;
  return PROQA_OK;
}

int proqa_topk_merge_gathered_device(const void* gathered_dev, int n_parts, int64_t nq, int k, uint32_t* status_host,
                                     float* D_dev, int64_t* I_dev, void* stream) {
  if (!gathered_dev || !D_dev || !I_dev || n_parts <= 0 || nq < 0 || k <= 0)
    return fail(PROQA_EINVAL, "topk_merge_gathered_device: bad argument");
  if ((long long)n_parts * k >= (1ll << 27))
    return fail(PROQA_EINVAL, "topk_merge_gathered_device: n_parts*k=%lld is too large", (long long)n_parts * k);
  size_t i_bytes = 0, d_bytes = 0, block = 0;
  if (int rc = proqa_sharded_block_layout(nq, k, &i_bytes, &d_bytes, &block)) return rc;
  const char* g = (const char*)gathered_dev;
  PROQA_HIP(launch_merge_lists((const float*)(g + i_bytes), (const long long*)g, n_parts, nq, k, (long long)(block / sizeof(float)),
                               (long long)(block / sizeof(int64_t)), D_dev, (long long*)I_dev, as_stream(stream),
                               (const unsigned*)(g + i_bytes + d_bytes), (long long)(block / sizeof(unsigned)), status_host));
  return PROQA_OK;
}

int proqa_sharded_search_device(proqa_index* idx, proqa_comm* c, const void* xq_dev, int64_t nq, int dtype, int k,
                                int64_t idx_offset, float* D_dev, int64_t* I_dev, void* stream) {
  if (!idx || !c || (nq > 0 && (!xq_dev || !D_dev || !I_dev)))
    return fail(PROQA_EINVAL, "sharded_search_device: NULL argument");
  if (nq < 0 || k <= 0) return fail(PROQA_EINVAL, "sharded_search_device: nq=%lld k=%d", (long long)nq, k);
  if (nq == 0) return PROQA_OK;   // every rank sees the same nq: nobody enters the collective
  if ((long long)c->world * k >= (1ll << 27))
    return fail(PROQA_EINVAL, "sharded_search_device: world_size*k=%lld is too large", (long long)c->world * k);
  int dev = 0;
  PROQA_HIP(hipGetDevice(&dev));
  if (dev != c->device) return fail(PROQA_EINVAL, "sharded_search_device: communicator lives on device %d, current is %d",
                                    c->device, dev);
  hipStream_t st = as_stream(stream);
  // one block per rank: [nq*k int64 ids | nq*k float scores | status word], each part padded to 16 bytes
  size_t i_bytes = 0, d_bytes = 0, block = 0;
  if (int rc = proqa_sharded_block_layout(nq, k, &i_bytes, &d_bytes, &block)) return rc;
  if (block > c->block_bytes) {
    // (the buffers may still be read by work an earlier call left on another stream: wait for the device, not for `st`)
    PROQA_HIP(hipDeviceSynchronize());
    if (c->send) PROQA_HIP(hipFree(c->send));
    if (c->recv) PROQA_HIP(hipFree(c->recv));
    c->send = c->recv = nullptr;
    c->block_bytes = 0;
    PROQA_HIP(hipMalloc((void**)&c->send, block));
    PROQA_HIP(hipMalloc((void**)&c->recv, block * c->world));
    c->block_bytes = block;
  }
  if (!c->status_host)
    PROQA_HIP(hipHostMalloc((void**)&c->status_host, (size_t)c->world * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
  uint32_t* status_dev = (uint32_t*)(c->send + i_bytes + d_bytes);
  // Local exact top-k with global ids, written straight into the send block.  The search is only ENQUEUED (its one host
  // synchronisation is deferred): the collective and the merge go onto the stream right behind it, and the host waits
  // once, at the end.  The status word travels with the block: 0 = this rank's list stands, 1 = one of its rounds
  // overflowed and the list will be rewritten by the overflow-safe re-scan, kPoison = this rank failed.  Every rank
  // sees every status, so all ranks agree on whether a second exchange is needed -- and a rank that fails still enters
  // the collective instead of leaving the others waiting in it.
  int rc_local = proqa_index_search_begin_device(idx, xq_dev, nq, dtype, k, idx_offset, (float*)(c->send + i_bytes),
                                                 (int64_t*)c->send, status_dev, stream);
  char local_error[512] = {0};
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (rc_local != PROQA_OK) {
      if (!local_error[0]) snprintf(local_error, sizeof local_error, "%s", proqa_last_error());
      PROQA_HIP(hipMemsetAsync(status_dev, 0xFF, sizeof(uint32_t), st));   // kPoison
    }
    // the ONE collective of the path: rank r's block lands at recv + r*block on every rank
    PROQA_RCCL(rccl().AllGather(c->send, c->recv, block, ncclChar, c->comm, st));
    // rank order == ascending row order, so the gathered position breaks score ties like the global row id
    // (the merge kernel also drops every rank's status word into pinned host memory: no copy command on the stream)
    PROQA_HIP(launch_merge_lists((const float*)(c->recv + i_bytes), (const long long*)c->recv, c->world, nq, k,
                                 (long long)(block / sizeof(float)), (long long)(block / sizeof(int64_t)), D_dev,
                                 (long long*)I_dev, st, (const unsigned*)(c->recv + i_bytes + d_bytes),
                                 (long long)(block / sizeof(unsigned)), c->status_host));
    int rewritten = 0;
    if (rc_local == PROQA_OK) {
      rc_local = proqa_index_search_finish(idx, &rewritten);   // waits for the stream; re-scans if a round overflowed
      if (rc_local != PROQA_OK) snprintf(local_error, sizeof local_error, "%s", proqa_last_error());
    }
    PROQA_HIP(hipStreamSynchronize(st));
    bool redo = false;
    for (int r = 0; r < c->world; ++r) {
      if (c->status_host[r] == kPoison) {
        if (local_error[0]) return fail(rc_local, "%s", local_error);
        return fail(PROQA_EHIP, "sharded_search_device: rank %d failed in its local search", r);
      }
      redo = redo || c->status_host[r] != 0;
    }
    if (!redo) {
      // (a finish that failed behind an all-zero exchange: the lists were exchanged, nobody waits, but this rank's
      // D / I may be incomplete -- report it instead of success)
      if (rc_local != PROQA_OK) return fail(rc_local, "%s", local_error);
      return PROQA_OK;
    }
    // some rank rewrote its list (here: `rewritten`): exchange and merge once more, with final lists
    if (rc_local == PROQA_OK) PROQA_HIP(hipMemsetAsync(status_dev, 0, sizeof(uint32_t), st));
    (void)rewritten;
  }
  return fail(PROQA_EHIP, "sharded_search_device: the ranks did not agree on a final result");
}

}  // extern "C"

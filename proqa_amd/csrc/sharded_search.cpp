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
  // one block per rank: [nq*k int64 ids | nq*k float scores], each part padded to 16 bytes
  const size_t i_bytes = round_up<size_t>((size_t)nq * k * sizeof(int64_t), 16);
  const size_t d_bytes = round_up<size_t>((size_t)nq * k * sizeof(float), 16);
  const size_t block = i_bytes + d_bytes;
  if (block > c->block_bytes) {
    PROQA_HIP(hipStreamSynchronize(st));
    if (c->send) PROQA_HIP(hipFree(c->send));
    if (c->recv) PROQA_HIP(hipFree(c->recv));
    c->send = c->recv = nullptr;
    c->block_bytes = 0;
    PROQA_HIP(hipMalloc((void**)&c->send, block));
    PROQA_HIP(hipMalloc((void**)&c->recv, block * c->world));
    c->block_bytes = block;
  }
  // local exact top-k with global ids, written straight into the send block
  if (int rc = proqa_index_search_device(idx, xq_dev, nq, dtype, k, idx_offset, (float*)(c->send + i_bytes),
                                         (int64_t*)c->send, stream))
    return rc;
  // the ONE collective of the path: rank r's block lands at recv + r*block on every rank
  PROQA_RCCL(rccl().AllGather(c->send, c->recv, block, ncclChar, c->comm, st));
  // rank order == ascending row order, so the gathered position breaks score ties like the global row id
  PROQA_HIP(launch_merge_lists((const float*)(c->recv + i_bytes), (const long long*)c->recv, c->world, nq, k,
                               (long long)(block / sizeof(float)), (long long)(block / sizeof(int64_t)), D_dev,
                               (long long*)I_dev, st));
  return PROQA_OK;
}

}  // extern "C"

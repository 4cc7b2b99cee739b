// Optional collective of the sharded 1-vs-N sweep for consumers that do not use torch.distributed.
//
// The sweep shards candidates over the ranks in contiguous blocks and needs NO data-path collective; the one exchange step is the
// gather of the (overlap f32, yaw i32) results -- 8 bytes per candidate -- to one rank.  RCCL (point-to-point over xGMI inside a
// node) is loaded on first use with dlopen: the library has no link-time dependency on it, and a process that already holds an
// RCCL (PyTorch ships its own) keeps using that one.  One send + one receive list inside a single group call: every rank sends
// its two arrays to `root`, the root posts world x 2 receives straight into the result arrays at the shard offsets.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "ovn_internal.h"

namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;
char g_rccl_why[512] = "";   // why loading failed, captured ONCE inside load_rccl (dlerror clears its message when read)

void load_rccl() {
  void* h = nullptr;
  // an RCCL that is already in the process first (RTLD_NOLOAD), then the ROCm installation's
  for (const char* name : {"librccl.so", "librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  if (!h)
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
  if (!h) {
    const char* why = dlerror();
    snprintf(g_rccl_why, sizeof(g_rccl_why), "%s", why ? why : "dlopen failed");
    return;
  }
  RcclApi a;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
  a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.GetErrorString;
  if (!a.ok) snprintf(g_rccl_why, sizeof(g_rccl_why), "a required ncclXxx symbol is missing");
  g_rccl = a;
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) {
    ovn_set_error("RCCL (librccl.so.1) could not be loaded: %s", g_rccl_why);
    return OVN_ERR_STATE;
  }
  return OVN_OK;
}

#define OVN_RCCL_CHECK(expr)                                                                        \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      ovn_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
      return OVN_ERR_HIP;                                                                           \
    }                                                                                               \
  } while (0)

// inside a ncclGroupStart/ncclGroupEnd bracket: close the group (ignoring its result) before reporting the error, so that the
// thread is not left inside an open group in which every later collective would hang
#define OVN_RCCL_CHECK_IN_GROUP(expr)                                                               \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      (void)g_rccl.GroupEnd();                                                                      \
      ovn_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
      return OVN_ERR_HIP;                                                                           \
    }                                                                                               \
  } while (0)

static_assert(sizeof(ncclUniqueId) == OVN_COMM_ID_BYTES, "OVN_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

}  // namespace

int ovn_comm_unique_id(unsigned char* id_out) {
  OVN_REQUIRE(id_out != nullptr, OVN_ERR_ARG, "ovn_comm_unique_id: id_out is NULL");
  int rc = need_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  OVN_RCCL_CHECK(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return OVN_OK;
}

int ovn_comm_init(ovn_ctx* ctx, int rank, int world_size, const unsigned char* id_bytes) {
  OVN_REQUIRE(ctx != nullptr && id_bytes != nullptr, OVN_ERR_ARG, "ovn_comm_init: NULL argument");
  OVN_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, OVN_ERR_ARG, "ovn_comm_init: rank %d of %d", rank, world_size);
  OVN_REQUIRE(ctx->comm == nullptr, OVN_ERR_STATE, "ovn_comm_init: the context already has a communicator");
  int rc = need_rccl();
  if (rc) return rc;
  OVN_ON_DEVICE(ctx->device);
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  OVN_RCCL_CHECK(g_rccl.CommInitRank(&comm, world_size, id, rank));
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world_size;
  return OVN_OK;
}

int ovn_comm_destroy(ovn_ctx* ctx) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_comm_destroy: ctx is NULL");
  if (!ctx->comm) return OVN_OK;
  OVN_ON_DEVICE(ctx->device);
  (void)hipDeviceSynchronize();
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
  ctx->comm = nullptr;
  OVN_RCCL_CHECK(g_rccl.CommDestroy(comm));
  return OVN_OK;
}

int ovn_gather_scores(ovn_ctx* ctx, const float* overlap_dev, const int32_t* yaw_dev, const int64_t* counts_host, int root,
                      float* overlap_all_dev, int32_t* yaw_all_dev, void* stream_) {
  OVN_REQUIRE(ctx != nullptr && ctx->comm != nullptr, OVN_ERR_STATE, "ovn_gather_scores: no communicator (ovn_comm_init)");
  OVN_REQUIRE(counts_host != nullptr, OVN_ERR_ARG, "ovn_gather_scores: counts is NULL");
  OVN_REQUIRE(root >= 0 && root < ctx->comm_world, OVN_ERR_ARG, "ovn_gather_scores: root %d of %d", root, ctx->comm_world);
  int64_t total = 0;
  for (int r = 0; r < ctx->comm_world; ++r) {
    OVN_REQUIRE(counts_host[r] >= 0 && counts_host[r] < (1ll << 31), OVN_ERR_ARG, "ovn_gather_scores: counts[%d] = %lld", r,
                (long long)counts_host[r]);
    total += counts_host[r];
  }
  const int64_t mine = counts_host[ctx->comm_rank];
  OVN_REQUIRE(mine == 0 || (overlap_dev && yaw_dev), OVN_ERR_ARG, "ovn_gather_scores: NULL shard buffer");
  OVN_REQUIRE(ctx->comm_rank != root || total == 0 || (overlap_all_dev && yaw_all_dev), OVN_ERR_ARG,
              "ovn_gather_scores: NULL result buffer on the root");
  OVN_ON_DEVICE(ctx->device);
  hipStream_t stream = (hipStream_t)stream_;
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
  OVN_RCCL_CHECK(g_rccl.GroupStart());
  if (mine > 0) {
    OVN_RCCL_CHECK_IN_GROUP(g_rccl.Send(overlap_dev, (size_t)mine, ncclFloat32, root, comm, stream));
    OVN_RCCL_CHECK_IN_GROUP(g_rccl.Send(yaw_dev, (size_t)mine, ncclInt32, root, comm, stream));
  }
  if (ctx->comm_rank == root) {
    int64_t off = 0;
    for (int r = 0; r < ctx->comm_world; ++r) {
      if (counts_host[r] > 0) {
        OVN_RCCL_CHECK_IN_GROUP(g_rccl.Recv(overlap_all_dev + off, (size_t)counts_host[r], ncclFloat32, r, comm, stream));
        OVN_RCCL_CHECK_IN_GROUP(g_rccl.Recv(yaw_all_dev + off, (size_t)counts_host[r], ncclInt32, r, comm, stream));
      }
      off += counts_host[r];
    }
  }
  OVN_RCCL_CHECK(g_rccl.GroupEnd());
  return OVN_OK;
}

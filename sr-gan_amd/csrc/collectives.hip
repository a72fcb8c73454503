// collectives.hip -- thin RCCL entry points of the C ABI ("comm init from a unique id, all-reduce fp32 /
// bf16 sum, tiny all-reduce for feature means"; reduce-scatter and all-gather for the point-to-point xGMI form, 8e).
//
// Host code only.  RCCL is resolved at FIRST USE with dlopen("librccl.so.1") -- RTLD_NOLOAD first, so a process that has
// imported PyTorch-ROCm binds to the copy torch already mapped (one RCCL per process, as with the HIP runtime) -- and
// the library carries no link-time dependency on it: a build box / CPU test that never calls these functions needs no
// RCCL.  One communicator = one (device, rank) pair, created on the CURRENT device; every collective is asynchronous on
// the caller's stream, in issue order.  No torch types, no global state besides the resolved function table.
#include <dlfcn.h>
#include <mutex>
#include <string.h>
#include "common.h"

namespace srgan {
namespace {

// the few RCCL declarations this file needs (rccl.h: NCCL_UNIQUE_ID_BYTES 128, ncclSum 0, ncclFloat32 7, ncclBfloat16 9)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int Result;
enum { NCCL_SUM = 0, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };

struct Rccl {
  void* handle = nullptr;
  Result (*GetUniqueId)(UniqueId*) = nullptr;
  Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  Result (*CommDestroy)(Comm) = nullptr;
  Result (*CommCount)(Comm, int*) = nullptr;
  Result (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  Result (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  Result (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(Result) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <typename F>
bool resolve(F& slot, const char* name) {
  slot = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  return slot != nullptr;
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);            // the copy this process already mapped, if any
      if (g_rccl.handle) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (g_rccl.handle) break;
      g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_rccl.handle) return;
    g_rccl.ok = resolve(g_rccl.GetUniqueId, "ncclGetUniqueId") && resolve(g_rccl.CommInitRank, "ncclCommInitRank") &&
                resolve(g_rccl.CommDestroy, "ncclCommDestroy") && resolve(g_rccl.CommCount, "ncclCommCount") &&
                resolve(g_rccl.AllReduce, "ncclAllReduce") && resolve(g_rccl.ReduceScatter, "ncclReduceScatter") &&
                resolve(g_rccl.AllGather, "ncclAllGather") && resolve(g_rccl.Broadcast, "ncclBroadcast") &&
                resolve(g_rccl.GetErrorString, "ncclGetErrorString");
  });
  return g_rccl.ok ? &g_rccl : nullptr;
}

// RCCL's result codes are small positive integers like hipError_t's; to keep the two apart in the ABI's "> 0 = hipError_t"
// convention a collective failure is reported as 10000 + ncclResult_t, with RCCL's own message in srgan_last_error().
int failed(const Rccl* r, Result result, const char* what) {
  set_error("%s failed: %s (ncclResult_t %d)", what, r->GetErrorString ? r->GetErrorString(result) : "?", result);
  return 10000 + result;
}

int wire_type(int32_t dtype) { return dtype == 0 ? NCCL_FLOAT32 : (dtype == 1 ? NCCL_BFLOAT16 : -1); }

#define SRGAN_RCCL(r, call, what)                   \
  do {                                              \
    const Result result_ = (call);                  \
    if (result_ != 0) return failed(r, result_, what); \
  } while (0)

}  // namespace
}  // namespace srgan

using namespace srgan;

extern "C" {

int srgan_comm_available(void) { return rccl() ? 1 : 0; }

int srgan_comm_unique_id(void* id128) {
  SRGAN_REQUIRE(id128, SRGAN_EINVAL, "srgan_comm_unique_id arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_comm_unique_id: librccl.so.1 could not be loaded");
  UniqueId id;
  SRGAN_RCCL(r, r->GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return SRGAN_OK;
}

int srgan_comm_init(void** comm, int32_t world_size, int32_t rank, const void* id128) {
  SRGAN_REQUIRE(comm && id128 && world_size >= 1 && rank >= 0 && rank < world_size, SRGAN_EINVAL, "srgan_comm_init arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_comm_init: librccl.so.1 could not be loaded");
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm created = nullptr;
  SRGAN_RCCL(r, r->CommInitRank(&created, world_size, id, rank), "ncclCommInitRank");
  *comm = created;
  return SRGAN_OK;
}

int srgan_comm_world_size(void* comm, int32_t* world_size) {
  SRGAN_REQUIRE(comm && world_size, SRGAN_EINVAL, "srgan_comm_world_size arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_comm_world_size: librccl.so.1 could not be loaded");
  int count = 0;
  SRGAN_RCCL(r, r->CommCount(comm, &count), "ncclCommCount");
  *world_size = count;
  return SRGAN_OK;
}

int srgan_comm_destroy(void* comm) {
  SRGAN_REQUIRE(comm, SRGAN_EINVAL, "srgan_comm_destroy arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_comm_destroy: librccl.so.1 could not be loaded");
  SRGAN_RCCL(r, r->CommDestroy(comm), "ncclCommDestroy");
  return SRGAN_OK;
}

int srgan_all_reduce_sum(void* comm, const void* send, void* recv, int64_t count, int32_t dtype, void* stream) {
  SRGAN_REQUIRE(comm && send && recv && count >= 0 && wire_type(dtype) >= 0, SRGAN_EINVAL, "srgan_all_reduce_sum arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_all_reduce_sum: librccl.so.1 could not be loaded");
  if (count == 0) return SRGAN_OK;
  SRGAN_RCCL(r, r->AllReduce(send, recv, (size_t)count, wire_type(dtype), NCCL_SUM, comm, (hipStream_t)stream), "ncclAllReduce");
  return SRGAN_OK;
}

int srgan_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recv_count, int32_t dtype, void* stream) {
  SRGAN_REQUIRE(comm && send && recv && recv_count >= 0 && wire_type(dtype) >= 0, SRGAN_EINVAL,
                "srgan_reduce_scatter_sum arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_reduce_scatter_sum: librccl.so.1 could not be loaded");
  if (recv_count == 0) return SRGAN_OK;
  SRGAN_RCCL(r, r->ReduceScatter(send, recv, (size_t)recv_count, wire_type(dtype), NCCL_SUM, comm, (hipStream_t)stream),
             "ncclReduceScatter");
  return SRGAN_OK;
}

int srgan_all_gather(void* comm, const void* send, void* recv, int64_t send_count, int32_t dtype, void* stream) {
  SRGAN_REQUIRE(comm && send && recv && send_count >= 0 && wire_type(dtype) >= 0, SRGAN_EINVAL, "srgan_all_gather arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_all_gather: librccl.so.1 could not be loaded");
  if (send_count == 0) return SRGAN_OK;
  SRGAN_RCCL(r, r->AllGather(send, recv, (size_t)send_count, wire_type(dtype), comm, (hipStream_t)stream), "ncclAllGather");
  return SRGAN_OK;
}

int srgan_broadcast(void* comm, void* buffer, int64_t count, int32_t dtype, int32_t root, void* stream) {
  SRGAN_REQUIRE(comm && buffer && count >= 0 && root >= 0 && wire_type(dtype) >= 0, SRGAN_EINVAL, "srgan_broadcast arguments");
  const Rccl* r = rccl();
  SRGAN_REQUIRE(r, SRGAN_EUNSUPPORTED, "srgan_broadcast: librccl.so.1 could not be loaded");
  if (count == 0) return SRGAN_OK;
  SRGAN_RCCL(r, r->Broadcast(buffer, buffer, (size_t)count, wire_type(dtype), root, comm, (hipStream_t)stream), "ncclBroadcast");
  return SRGAN_OK;
}

}  // extern "C"

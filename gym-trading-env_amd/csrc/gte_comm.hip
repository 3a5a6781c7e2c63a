// gte_comm.hip — the ONE exchange step of the sharded path: the RCCL all-gather of the per-step
// returns (reward f32 | terminated u8 | truncated u8 = 6 bytes per env, which the step kernel
// writes in exactly that packed layout) and, on request, of the observations, called from inside
// the library so that a C caller of libgte has a multi-GPU path and a per-step gather costs no
// framework overhead (SURVEY §8b `gte_allgather`, §8e).
//
// RCCL is bound at run time (dlopen of the librccl already in the process — PyTorch's when the
// host layer is Python — else the system one): libgte.so itself has no link-time dependency on
// it and loads on hosts without RCCL.  One communicator per env, created from an id the caller
// distributes (rank 0: gte_comm_unique_id; every rank: gte_comm_init).
//
// Modes of a gather (gte.h): 0 = on the env's stream, i.e. stream-ordered between two steps
// (the synchronous per-step form: step t -> gather t -> step t+1, no host synchronisation);
// 1 = on the library's communication stream behind an event recorded on the env's stream, so
// that the steps enqueued afterwards overlap it (the caller rotates return buffers with
// gte_bind_returns / gathers whole blocks of rows, and joins with gte_comm_wait).
//
// On the fully connected 7-link xGMI topology the direct (one-shot) all-gather moves each
// shard over all links at once, where a ring is bound by one link per hop; which
// one RCCL uses is RCCL's choice, steerable from outside with NCCL_ALGO / NCCL_PROTO
//  — the payload here is 6 bytes per env, far below either limit.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "gte_device.h"

namespace gte {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

static RcclApi& rccl() {
  static RcclApi api;
  return api;
}

// -> nullptr on success, else why RCCL is not usable
const char* rccl_load() {
  RcclApi& a = rccl();
  if (a.AllGather) return nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  // a copy already loaded into the process first (two RCCLs in one process would each keep
  // their own view of the devices), then a fresh load
  for (int pass = 0; pass < 2 && !a.handle; ++pass)
    for (const char* n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (a.handle) break;
    }
  if (!a.handle) a.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!a.handle) {
    a.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
    return a.error.c_str();
  }
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
  a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !a.AllGather) {
    a.AllGather = nullptr;
    a.error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / "
              "ncclCommDestroy / ncclGetErrorString";
    return a.error.c_str();
  }
  return nullptr;
}

const char* rccl_error(int code) { return rccl().GetErrorString((ncclResult_t)code); }

int rccl_unique_id(uint8_t* out128) {
  ncclUniqueId id;
  const ncclResult_t r = rccl().GetUniqueId(&id);
  if (r == ncclSuccess) memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return (int)r;
}

int rccl_comm_init(void** comm, const uint8_t* id128, int rank, int world) {
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t r = rccl().CommInitRank(&c, world, id, rank);
  *comm = (void*)c;
  return (int)r;
}

int rccl_allgather_bytes(void* comm, const void* src, void* dst, size_t bytes, hipStream_t stream) {
  return (int)rccl().AllGather(src, dst, bytes, ncclUint8, (ncclComm_t)comm, stream);
}

int rccl_comm_destroy(void* comm) { return (int)rccl().CommDestroy((ncclComm_t)comm); }

}  // namespace gte

// ba_comm.h -- RCCL communicator for the sharded ChainBundle solve.
//
// The reduced pose system, the robust-statistics histograms and a handful of scalars are summed across
// the ranks of one node with ncclAllReduce (RCCL over xGMI), enqueued on the solver's own HIP stream, so
// the collective is stream-ordered with the kernels around it (no host round trip).  librccl is opened
// lazily with dlopen: a single-GPU user never needs it.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>

namespace mcp {

struct RcclUniqueId { char internal[128]; };

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (lib) return true;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) return false;
    GetUniqueId = (int (*)(RcclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(lib, "ncclCommInitRank");
    AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
    CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    return GetUniqueId && CommInitRank && AllReduce && CommDestroy;
  }
};
inline RcclApi& rccl() { static RcclApi api; return api; }

constexpr int RCCL_FLOAT64 = 8;   // ncclFloat64
constexpr int RCCL_SUM = 0;       // ncclSum

}  // namespace mcp

struct mcp_comm { void* comm = nullptr; int rank = 0, world = 1, device = 0; };

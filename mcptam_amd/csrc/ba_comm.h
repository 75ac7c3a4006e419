// ba_comm.h -- RCCL communicator for the sharded ChainBundle solve.
//
// The reduced pose system, the robust-statistics histograms and a handful of scalars are summed across
// the ranks of one node with ncclAllReduce (RCCL over xGMI), enqueued on the solver's own HIP stream, so
// the collective is stream-ordered with the kernels around it (no host round trip).  librccl is opened
// lazily with dlopen: a single-GPU user never needs it.
//
// A communicator carries TWO RCCL communicators over the same ranks ("lanes"): lane 0 serves the solver's main stream (the
// system the trial in hand needs, its result scalars, the median), lane 1 the speculative stream (the systems of the next
// lambdas and the trials evaluated ahead).  One RCCL communicator executes its operations strictly in enqueue order, so two
// streams sharing one would serialise on each other; with a lane per stream the speculative chain's collectives never sit in
// front of the trial's.  Every rank enqueues the same sequence per lane (the host control flow depends only on all-reduced
// values), which is the ordering RCCL requires.  Lane 1 is derived from lane 0 with ncclCommSplit (same colour on all ranks).
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
  int (*CommSplit)(void*, int, int, void**, void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*CommGetAsyncError)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (lib) return true;
    // a process that has imported torch already maps torch's bundled librccl (soname librccl.so.1): take THAT copy rather
    // than a second RCCL instance with its own bootstrap threads and IPC state (RTLD_NOLOAD resolves by soname among the
    // loaded objects); only a process without one loads the system library
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }
    if (!lib) for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) return false;
    GetUniqueId = (int (*)(RcclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(lib, "ncclCommInitRank");
    AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
    CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    CommSplit = (int (*)(void*, int, int, void**, void*))dlsym(lib, "ncclCommSplit");
    CommAbort = (int (*)(void*))dlsym(lib, "ncclCommAbort");
    CommGetAsyncError = (int (*)(void*, int*))dlsym(lib, "ncclCommGetAsyncError");
    GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    return GetUniqueId && CommInitRank && AllReduce && CommDestroy;
  }
};
inline RcclApi& rccl() { static RcclApi api; return api; }

constexpr int RCCL_FLOAT64 = 8;   // ncclFloat64
constexpr int RCCL_SUM = 0;       // ncclSum

}  // namespace mcp

struct mcp_comm {
  void* comm = nullptr;        // lane 0: main stream
  void* comm2 = nullptr;       // lane 1: speculative stream (ncclCommSplit of lane 0)
  int rank = 0, world = 1, device = 0;
  bool dead = false;           // aborted by the watchdog: no further collective may be enqueued
  void* lane(int l) const { return (l && comm2) ? comm2 : comm; }
};

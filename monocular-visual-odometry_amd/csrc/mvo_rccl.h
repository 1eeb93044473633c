// mvo_rccl.h -- RCCL, bound at run time.
//
// The observation-partitioned solve all-reduces 8 fp64 per frame pair over
// xGMI between the observation pass and the (identical) LM step of every rank.  librccl.so carries
// ~570 MB of kernels; a single-GPU user of libmvo_hip (the reference's own use) should not pay for
// loading it, so it is opened on the first mvo_hip_comm_* call (dlopen "librccl.so.1": the same
// search and SONAME rules as a DT_NEEDED entry -- in a process that already holds an RCCL, e.g.
// PyTorch's, that copy and its HIP runtime are the ones used, as for libamdhip64).  Types come from
// <rccl/rccl.h>; only the entry points below are resolved.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

struct MvoRccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: mvo_hip_comm_abort falls back on CommDestroy
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

inline const MvoRccl& mvo_rccl() {
  static MvoRccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(lib, "ncclCommAbort"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
  });
  return r;
}

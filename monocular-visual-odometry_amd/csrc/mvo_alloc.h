// mvo_alloc.h -- the one place libmvo_hip.so allocates device memory.
//
// hipMalloc hands out memory with whatever it last held.  With MVO_HIP_POISON_ALLOC set in the
// environment (the GPU tests set it, tests/conftest.py) every allocation is filled with 0xFF bytes
// first -- NaN as a double, 0xFFFFFFFF as an index -- so that a kernel reading device memory nobody
// wrote shows up as a wrong answer or a refused index instead of passing on a fresh box's zeros.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

inline hipError_t mvo_device_alloc(void** p, size_t bytes) {
  static const bool poison = std::getenv("MVO_HIP_POISON_ALLOC") != nullptr;
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && poison) {
    // hipMemset runs on the null stream and may return before it is done; the library's own streams
    // are non-blocking ones that do not wait for it
    e = hipMemset(*p, 0xFF, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  return e;
}

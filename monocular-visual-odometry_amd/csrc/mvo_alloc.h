// mvo_alloc.h -- the one place libmvo_hip.so allocates and frees device memory (defined in
// mvo_hip.hip, shared by every translation unit of the library).
//
// hipMalloc hands out memory with whatever it last held, and a store past the end of an allocation
// usually lands in a neighbour without any fault.  With MVO_HIP_POISON_ALLOC set in the environment
// (the GPU tests set it, tests/conftest.py) every allocation
//   * is filled with 0xFF bytes first -- NaN as a double, 0xFFFFFFFF as an index -- so that a kernel
//     reading device memory nobody wrote shows up as a wrong answer instead of passing on a fresh
//     box's zeros, and
//   * carries a 256-byte guard behind its end, checked when it is freed and by
//     mvo_hip_debug_check_guards() (include/mvo_hip.h) at any time.
// Unset (production) the two functions are hipMalloc / hipFree.
#pragma once
#include <hip/hip_runtime.h>

hipError_t mvo_device_alloc(void** p, size_t bytes);
hipError_t mvo_device_free(void* p);

// mvo_grid.h -- the one cross-workgroup wait of libmvo_hip: a counting barrier for a grid whose
// workgroups are ALL resident at once (at most one per CU; the host checks), used by the persistent robust
// solver (k_lm<256, 6>, mvo_lm.h).  Validated stand-alone by scripts/exp/grid_barrier_probe.hip.
//
// Protocol (gfx950: 8 XCDs, one L2 each, coherent with each other only through write-through stores and
// device-scope loads):
//   * what a workgroup hands to the others it stores device-coherently (__hip_atomic_store, agent scope:
//     written through its XCD's L2) BEFORE it calls grid_sync;
//   * grid_sync: every wavefront waits for its own stores (s_waitcnt vmcnt(0) -- a workgroup-scope fence
//     does not, see block_store_partials in mvo_kernels.h), workgroup barrier, thread 0 adds 1 to the
//     counter (agent scope) and polls it until it reads n_wg * epoch, workgroup barrier;
//   * what the others stored is then read with device-scope loads (__hip_atomic_load, agent scope).
// The counter only counts up (epoch k waits for n_wg * k): nothing is reset inside the kernel, a workgroup
// that is one barrier ahead cannot disturb one that is still reading the count.
// EVERY WAIT IS BOUNDED.  A workgroup that has polled kGridSpinLimit times gives up, raises the error word
// and returns false; every poll also looks at the error word, so one workgroup giving up ends the waits of
// all the others within one poll.  The caller leaves its loop on `false`: the grid always drains, whatever
// went wrong (a workgroup that never became resident, diverged control flow), and the host sees the error
// word and falls back on the launch-per-evaluation path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct GridSync {
  uint32_t* count;  // zeroed by the host before the launch
  uint32_t* error;  // zeroed by the host before the launch; != 0 afterwards: some workgroup gave up
};

constexpr uint32_t kGridSpinLimit = 1u << 21;  // x (s_sleep 1 + two L2 round trips ~ 1 us) ~ 2 s

__device__ __forceinline__ bool grid_sync(const GridSync& gs, uint32_t n_wg, uint32_t& epoch) {
  __shared__ uint32_t grid_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's coherent stores have been acknowledged
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    const uint32_t target = n_wg * epoch;
    __hip_atomic_fetch_add(gs.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t ok = 1;
    for (uint32_t spins = 0;; ++spins) {
      if (__hip_atomic_load(gs.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      if (spins >= kGridSpinLimit || __hip_atomic_load(gs.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(gs.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    grid_ok = ok;
  }
  __syncthreads();
  return grid_ok != 0;
}

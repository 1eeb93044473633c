// Probe for the hand-shake of a persistent multi-workgroup kernel on gfx950 (8 XCDs, one L2 each):
//   every workgroup stores data with device-coherent (sc1, write-through) stores, waits for them
//   (s_waitcnt vmcnt(0) in every wavefront), arrives at a counter; once all have arrived every workgroup
//   reads ALL workgroups' data with device-coherent loads and checks it.  Two data buffers, used in turn.
// Every wait is bounded: a workgroup that gives up raises a flag that ends all the others too, so the
// grid always drains.  Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe.bin grid_barrier_probe.hip
// Run: ./grid_barrier_probe.bin [workgroups] [epochs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../monocular-visual-odometry_amd/csrc/mvo_grid.h"

__global__ __launch_bounds__(256) void k_probe(GridSync gs, double* data, uint32_t n_wg, uint32_t epochs,
                                               unsigned long long* mismatches, unsigned long long* cycles) {
  uint32_t epoch = 0;
  const uint32_t wg = blockIdx.x;
  unsigned long long bad = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t e = 0; e < epochs; ++e) {
    double* buf = data + (size_t)(e & 1) * n_wg * 8;
    if (threadIdx.x < 8)
      __hip_atomic_store(buf + (size_t)wg * 8 + threadIdx.x, (double)(e * 1000 + wg) + 0.125 * threadIdx.x, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_sync(gs, n_wg, epoch)) break;  // (uniform: every thread of the workgroup gets the same answer)
    for (uint32_t i = threadIdx.x; i < n_wg * 8; i += 256) {
      const double v = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (double)(e * 1000 + (i >> 3)) + 0.125 * (i & 7)) ++bad;
    }
  }
  if (bad) atomicAdd(mismatches, bad);
  if (wg == 0 && threadIdx.x == 0) *cycles = __builtin_amdgcn_s_memtime() - t0;
}

int main(int argc, char** argv) {
  const uint32_t n_wg = argc > 1 ? (uint32_t)atoi(argv[1]) : 116, epochs = argc > 2 ? (uint32_t)atoi(argv[2]) : 2000;
  if (n_wg == 0 || n_wg > 256) { fprintf(stderr, "1..256 workgroups (all must be resident at once)\n"); return 2; }
  uint32_t* ctl;
  double* data;
  unsigned long long *mm, h[2] = {0, 0};
  if (hipMalloc(&ctl, 256) != hipSuccess || hipMalloc(&data, sizeof(double) * 2 * n_wg * 8) != hipSuccess ||
      hipMalloc(&mm, 16) != hipSuccess)
    return 3;
  hipMemset(ctl, 0, 256);
  hipMemset(data, 0xFF, sizeof(double) * 2 * n_wg * 8);
  hipMemset(mm, 0, 16);
  GridSync gs;
  gs.count = ctl;
  gs.error = ctl + 32;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  k_probe<<<n_wg, 256>>>(gs, data, n_wg, epochs, mm, mm + 1);
  hipEventRecord(b);
  const hipError_t rc = hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  uint32_t hc[64];
  hipMemcpy(hc, ctl, 256, hipMemcpyDeviceToHost);
  hipMemcpy(h, mm, 16, hipMemcpyDeviceToHost);
  printf("workgroups %u epochs %u: rc %d, arrivals %u (expected %u), gave up %u, mismatches %llu, %.3f us per epoch (%.0f cycles)\n",
         n_wg, epochs, (int)rc, hc[0], n_wg * epochs, hc[32], h[0], 1e3 * ms / epochs, (double)h[1] / epochs);
  return (rc == hipSuccess && hc[32] == 0 && h[0] == 0 && hc[0] == n_wg * epochs) ? 0 : 1;
}

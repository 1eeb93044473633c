// Dependent-issue latency and issue rate of v_fma_f64 (and v_rcp_f64, LDS round trip) for ONE wavefront.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/exp/fma_latency.bin scripts/exp/fma_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k_fma(double* out, unsigned long long* cyc, double a, double b) {
  double v[CHAINS];
  for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x + c;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) v[c] = __builtin_fma(v[c], a, b);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += v[c];
  out[threadIdx.x] = s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcp(double* out, unsigned long long* cyc, double a) {
  double v = threadIdx.x + 1.5;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v = __builtin_amdgcn_rcp(v) + a;
  }
  out[threadIdx.x] = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, unsigned long long* cyc) {
  __shared__ int idx[256];
  for (int i = threadIdx.x; i < 256; i += 64) idx[i] = (i * 7 + 1) & 255;
  __syncthreads();
  int p = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) p = idx[p];
  }
  out[threadIdx.x] = p;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
  auto report = [&](const char* name, double n) {
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %6.1f s_memtime ticks per op\n", name, (double)h / n);
  };
  for (int rep = 0; rep < 2; ++rep) k_fma<1><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 1 chain", 1024.0);
  for (int rep = 0; rep < 2; ++rep) k_fma<2><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 2 chains (per fma)", 2048.0);
  for (int rep = 0; rep < 2; ++rep) k_fma<4><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 4 chains (per fma)", 4096.0);
  for (int rep = 0; rep < 2; ++rep) k_fma<8><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 8 chains (per fma)", 8192.0);
  for (int rep = 0; rep < 2; ++rep) k_rcp<<<1, 64>>>(out, cyc, 0.25);
  report("v_rcp_f64 + add, 1 chain", 1024.0);
  for (int rep = 0; rep < 2; ++rep) k_lds<<<1, 64>>>(out, cyc);
  report("LDS dependent read", 1024.0);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("shader clock %d kHz, wall clock %d kHz\n", clk, wall);
  return 0;
}

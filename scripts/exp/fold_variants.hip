// Experiment (not part of the library): how fast can the fold's input be read?
//   A  current pattern: 8-byte loads, 24-byte lane stride (AoS rays), 6 loads per residual block
//   B  16-byte loads, one lane = two consecutive blocks (48 contiguous bytes per array)
//   C  ceiling: the same bytes as plain contiguous 16-byte loads (sum only)
//   D  LDS staged: contiguous 16-byte loads -> LDS -> stride-3 reads
// Build: hipcc --offload-arch=gfx950 -O3 -o fold_variants scripts/exp/fold_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double wsum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

__device__ __forceinline__ void accum(double x0, double y0, double z0, double x1, double y1, double z1, double* S) {
  const double vx = y0 * z1 - z0 * y1, vy = z0 * x1 - x0 * z1, vz = x0 * y1 - y0 * x1;
  S[0] += vx * vx; S[1] += vx * vy; S[2] += vx * vz; S[3] += vy * vy; S[4] += vy * vz; S[5] += vz * vz;
}

__global__ __launch_bounds__(256) void kA(const double* m0, const double* m1, size_t n, size_t chunk, double* out) {
  const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
  const size_t lo = gw * chunk, hi = lo + chunk < n ? lo + chunk : n;
  double S[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (size_t i = lo + lane; i < hi; i += 64)
    accum(m0[3 * i], m0[3 * i + 1], m0[3 * i + 2], m1[3 * i], m1[3 * i + 1], m1[3 * i + 2], S);
  double t = 0; for (int k = 0; k < 6; ++k) t += wsum(S[k]);
  if (lane == 0) out[gw] = t;
}

__global__ __launch_bounds__(256) void kB(const double* m0, const double* m1, size_t n, size_t chunk, double* out) {
  const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
  const size_t lo = gw * chunk, hi = lo + chunk < n ? lo + chunk : n;  // chunk even
  double S[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 2
  for (size_t i = lo + 2 * lane; i + 1 < hi; i += 128) {
    const double2* a = reinterpret_cast<const double2*>(m0 + 3 * i);
    const double2* b = reinterpret_cast<const double2*>(m1 + 3 * i);
    const double2 a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
    accum(a0.x, a0.y, a1.x, b0.x, b0.y, b1.x, S);
    accum(a1.y, a2.x, a2.y, b1.y, b2.x, b2.y, S);
  }
  double t = 0; for (int k = 0; k < 6; ++k) t += wsum(S[k]);
  if (lane == 0) out[gw] = t;
}

__global__ __launch_bounds__(256) void kC(const double* m0, const double* m1, size_t n, size_t chunk, double* out) {
  const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
  const size_t lo = gw * chunk * 3, hi = (gw * chunk + chunk < n ? gw * chunk + chunk : n) * 3;  // doubles
  double s0 = 0, s1 = 0;
#pragma unroll 4
  for (size_t i = lo + 2 * lane; i + 1 < hi; i += 128) {
    const double2 a = *reinterpret_cast<const double2*>(m0 + i), b = *reinterpret_cast<const double2*>(m1 + i);
    s0 += a.x * b.x; s1 += a.y * b.y;
  }
  const double t = wsum(s0 + s1);
  if (lane == 0) out[gw] = t;
}

// one wave stages 128 blocks (2 x 3 KB) per step through its own LDS slice
__global__ __launch_bounds__(256) void kD(const double* m0, const double* m1, size_t n, size_t chunk, double* out) {
  __shared__ double lds[4][2][384];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gw = (size_t)blockIdx.x * 4 + wv;
  const size_t lo = gw * chunk, hi = lo + chunk < n ? lo + chunk : n;  // chunk multiple of 128
  double S[6] = {0, 0, 0, 0, 0, 0};
  for (size_t base = lo; base + 128 <= hi; base += 128) {
    const double2* a = reinterpret_cast<const double2*>(m0 + 3 * base);
    const double2* b = reinterpret_cast<const double2*>(m1 + 3 * base);
    double2 ra[3], rb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ra[k] = a[lane + 64 * k]; rb[k] = b[lane + 64 * k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      *reinterpret_cast<double2*>(&lds[wv][0][2 * (lane + 64 * k)]) = ra[k];
      *reinterpret_cast<double2*>(&lds[wv][1][2 * (lane + 64 * k)]) = rb[k];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double* pa = &lds[wv][0][3 * (lane + 64 * h)];
      const double* pb = &lds[wv][1][3 * (lane + 64 * h)];
      accum(pa[0], pa[1], pa[2], pb[0], pb[1], pb[2], S);
    }
    __builtin_amdgcn_wave_barrier();
  }
  double t = 0; for (int k = 0; k < 6; ++k) t += wsum(S[k]);
  if (lane == 0) out[gw] = t;
}

int main(int argc, char** argv) {
  // ~config 4 (multiple of 128 * 4096); `fold_variants 4` = four times that: ~config 3, 1.9 GB
  const size_t n = 9830400 * (size_t)(argc > 1 ? atoi(argv[1]) : 1);
  const size_t waves = 4096, chunk = n / waves;
  std::vector<double> h(3 * n);
  for (size_t i = 0; i < 3 * n; ++i) h[i] = 0.001 * (double)((i * 2654435761u) % 1000) - 0.5;
  double *m0, *m1, *out;
  CK(hipMalloc(&m0, 24 * n)); CK(hipMalloc(&m1, 24 * n)); CK(hipMalloc(&out, 8 * waves));
  CK(hipMemcpy(m0, h.data(), 24 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(m1, h.data(), 24 * n, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"A 8B strided", "B 16B two-per-lane", "C contiguous ceiling", "D LDS staged"};
  for (int round = 0; round < 3; ++round)
    for (int v = 0; v < 4; ++v) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(kA, dim3(waves / 4), dim3(256), 0, 0, m0, m1, n, chunk, out);
        if (v == 1) hipLaunchKernelGGL(kB, dim3(waves / 4), dim3(256), 0, 0, m0, m1, n, chunk, out);
        if (v == 2) hipLaunchKernelGGL(kC, dim3(waves / 4), dim3(256), 0, 0, m0, m1, n, chunk, out);
        if (v == 3) hipLaunchKernelGGL(kD, dim3(waves / 4), dim3(256), 0, 0, m0, m1, n, chunk, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      std::vector<double> o(waves); CK(hipMemcpy(o.data(), out, 8 * waves, hipMemcpyDeviceToHost));
      double chk = 0; for (double x : o) chk += x;
      printf("round %d  %-22s %8.1f us  %7.1f GB/s  (checksum %.6e)\n", round, names[v], best * 1e3, 48.0 * n / (best * 1e-3) / 1e9, chk);
    }
  return 0;
}

// v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950, for the question VERDICT (round 1) item 9 asks:
// would the trailing rank-6 update of k_ba_solve's 168 x 168 block L D L^T (and k_ba_schur's Y E^T) be
// faster on the matrix cores?
//   (1) issue rate / latency of the instruction itself, one wavefront and two wavefronts per SIMD
//   (2) the trailing update of ONE block column sweep (28 block columns, shrinking active set, one
//       barrier per column, operands published in LDS as k_ba_solve does), thread-per-6x6-block on the
//       VALU (k_ba_solve's step 3 as it stands) against 16 x 16 MFMA tiles (K = 6 padded to 8)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/exp/mfma_f64_probe.bin scripts/exp/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(1024) void k_mfma(double* out, unsigned long long* cyc, double a, double b) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  const double av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[threadIdx.x] = s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the time of the LAST wavefront: the oldest one is served first and alone would read as unslowed
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS>
__global__ __launch_bounds__(1024) void k_fma(double* out, unsigned long long* cyc, double a, double b) {
  double v[CHAINS];
  for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x + c;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
  __syncthreads();  // the time of the LAST wavefront: the oldest one is served first and alone would read as unslowed
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// tick calibration: a long dependent chain timed by s_memtime inside and by HIP events outside
__global__ void k_spin(double* out, unsigned long long* cyc, double a, double b, int iters) {
  double v = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v = __builtin_fma(v, a, b);
  }
  out[threadIdx.x] = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- (2) the trailing update, VALU: thread <-> block (ib, jb) of the lower block triangle, jb <= ib <= M
// (block row M = the right-hand side), enumerated column by column as in k_ba_solve
#define MB 28
#define NBLK ((MB + 1) * (MB + 2) / 2)
__global__ __launch_bounds__(512) void k_trail_valu(double* out, unsigned long long* cyc, const double* __restrict__ seed) {
  __shared__ __align__(16) double pubX[MB + 1][38];
  __shared__ double dinvs[6];
  const unsigned tid = threadIdx.x;
  unsigned jb = 0;
  for (unsigned q = 1; q <= MB; ++q)
    if (tid >= q * (MB + 1) - (q * (q - 1)) / 2) jb = q;
  const unsigned ib = jb + (tid - (jb * (MB + 1) - (jb * (jb - 1)) / 2));
  const bool have = tid < NBLK;
  double a[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) a[e] = seed[(tid * 36 + e) & 1023];
  for (unsigned i = tid; i < (MB + 1) * 38; i += 512) (&pubX[0][0])[i] = seed[i & 1023] * 1e-3;
  if (tid < 6) dinvs[tid] = 1.0 + 0.01 * tid;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (unsigned kb = 0; kb < MB; ++kb) {
    if (have && jb > kb) {
      double x[36], wv[36], di[6];
      const double2* __restrict__ xj = reinterpret_cast<const double2*>(pubX[jb]);
      const double2* __restrict__ xi = reinterpret_cast<const double2*>(pubX[ib]);
#pragma unroll
      for (int c = 0; c < 6; ++c) di[c] = dinvs[c];
#pragma unroll
      for (int e = 0; e < 18; ++e) {
        const double2 vj = xj[e], vi = xi[e];
        x[2 * e] = vj.x;
        x[2 * e + 1] = vj.y;
        wv[2 * e] = vi.x * di[(2 * e) % 6];
        wv[2 * e + 1] = vi.y * di[(2 * e + 1) % 6];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double acc = a[6 * r + c];
#pragma unroll
          for (int p = 0; p < 6; ++p) acc -= wv[6 * r + p] * x[6 * c + p];
          a[6 * r + c] = acc;
        }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int e = 0; e < 36; ++e) s += a[e];
  out[tid] = s;
  if (tid == 0) cyc[0] = t1 - t0;
}

// ---- (2) the trailing update, MFMA: the (6 MB + 6) x 6 MB lower triangle in 16 x 16 tiles (11 x 11 grid,
// 66 lower tiles, tile t of wavefront w = w + 8 t'), 4 accumulators per lane and tile.  Per block column:
// A operand = W rows of the tile row (K 0..3 and K 4..7, entries 6 and 7 zero), B operand = X rows of the
// tile column.  Tiles wholly above/left of the active region are skipped (wave-uniform).
#define NT16 11
#define NTILES (NT16 * (NT16 + 1) / 2)
#define TPW ((NTILES + 7) / 8)
__global__ __launch_bounds__(512) void k_trail_mfma(double* out, unsigned long long* cyc, const double* __restrict__ seed) {
  __shared__ double Wp[NT16 * 16][9], Xp[NT16 * 16][9];  // 9: the 16 rows a quarter-wave reads fall in distinct banks
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (unsigned i = tid; i < NT16 * 16 * 9; i += 512) {
    const bool pad = (i % 9) >= 6;
    (&Wp[0][0])[i] = pad ? 0.0 : seed[i & 1023] * 1e-3;
    (&Xp[0][0])[i] = pad ? 0.0 : seed[(i + 7) & 1023] * 1e-3;
  }
  d4 acc[TPW];
  int ti[TPW], tj[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int t = (int)wave + 8 * q;
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    ti[q] = t < NTILES ? i : -1;
    tj[q] = t - i * (i + 1) / 2;
    acc[q] = d4{seed[lane], seed[lane + 64], seed[lane + 128], seed[lane + 192]};
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (unsigned kb = 0; kb < MB; ++kb) {
    const int first = 6 * ((int)kb + 1);  // first row / column still active
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      if (ti[q] < 0 || 16 * tj[q] + 15 < first) continue;  // wave-uniform (ti >= tj: the row test follows)
      const double a0 = Wp[16 * ti[q] + (lane & 15)][lane >> 4], a1 = Wp[16 * ti[q] + (lane & 15)][4 + (lane >> 4)];
      const double b0 = Xp[16 * tj[q] + (lane & 15)][lane >> 4], b1 = Xp[16 * tj[q] + (lane & 15)][4 + (lane >> 4)];
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[q], 0, 0, 0);
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int q = 0; q < TPW; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[tid] = s;
  if (tid == 0) cyc[0] = t1 - t0;
}

int main() {
  double *out, *seed;
  unsigned long long* cyc;
  unsigned long long h;
  if (hipMalloc(&out, 8 * 1024) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess || hipMalloc(&seed, 8 * 1024) != hipSuccess) return 1;
  double hs[1024];
  for (int i = 0; i < 1024; ++i) hs[i] = 0.5 + (double)((i * 37) % 101) / 101.0;
  hipMemcpy(seed, hs, sizeof hs, hipMemcpyHostToDevice);
  auto report = [&](const char* name, double n) {
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.1f ticks per op  (%llu ticks)\n", name, (double)h / n, h);
  };
  for (int rep = 0; rep < 2; ++rep) k_mfma<1><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("mfma f64 16x16x4, 1 wave, 1 chain (latency)", 1024.0);
  for (int rep = 0; rep < 2; ++rep) k_mfma<4><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("mfma f64 16x16x4, 1 wave, 4 chains (per mfma)", 4096.0);
  for (int rep = 0; rep < 2; ++rep) k_mfma<4><<<1, 256>>>(out, cyc, 0.999, 0.5);
  report("mfma f64 16x16x4, 4 waves (1/SIMD), 4 chains", 4096.0);
  for (int rep = 0; rep < 2; ++rep) k_mfma<4><<<1, 512>>>(out, cyc, 0.999, 0.5);
  report("mfma f64 16x16x4, 8 waves (2/SIMD), 4 chains", 4096.0);
  for (int rep = 0; rep < 2; ++rep) k_mfma<4><<<1, 1024>>>(out, cyc, 0.999, 0.5);
  report("mfma f64 16x16x4, 16 waves (4/SIMD), 4 chains", 4096.0);
  for (int rep = 0; rep < 2; ++rep) k_fma<8><<<1, 64>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 1 wave, 8 chains (per fma)", 8192.0);
  for (int rep = 0; rep < 2; ++rep) k_fma<8><<<1, 512>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 8 waves (2/SIMD), 8 chains (per fma & wave)", 8192.0);
  for (int rep = 0; rep < 2; ++rep) k_fma<8><<<1, 1024>>>(out, cyc, 0.999, 0.5);
  report("fma f64, 16 waves (4/SIMD), 8 chains (per fma & wave)", 8192.0);
  for (int rep = 0; rep < 3; ++rep) k_trail_valu<<<1, 512>>>(out, cyc, seed);
  report("trailing update, 28 columns, VALU blocks (per col)", 28.0);
  for (int rep = 0; rep < 3; ++rep) k_trail_mfma<<<1, 512>>>(out, cyc, seed);
  report("trailing update, 28 columns, MFMA tiles (per col)", 28.0);
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_spin<<<1, 64>>>(out, cyc, 0.999, 0.5, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    k_spin<<<1, 64>>>(out, cyc, 0.999, 0.5, 200000);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %llu ticks in %.3f ms by events -> %.3f ns per tick; dependent fma %.2f ticks\n", h, ms,
           ms * 1e6 / (double)h, (double)h / (200000.0 * 16.0));
  }
  int clk = 0, wall = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("shader clock %d kHz, s_memtime clock %d kHz\n", clk, wall);
  return 0;
}

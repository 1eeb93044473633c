// Probe: relative error of the raw v_rcp_f64 / v_rsq_f64 seeds on gfx950 and after 1 / 2 refinement steps,
// and of a 13-term exp against libm.  Build: hipcc --offload-arch=gfx950 -O3 -o rcp_precision rcp_precision.hip
#include <hip/hip_runtime.h>
#include "../../monocular-visual-odometry_amd/csrc/mvo_device.h"
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  o[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  o[n + i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  o[2 * n + i] = r;
  double y = __builtin_amdgcn_rsq(v);
  o[3 * n + i] = y;
  double g = v * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  o[4 * n + i] = h + h;
  e = fma(-h, g, 0.5);
  h = fma(h, e, h);
  o[5 * n + i] = h + h;
}

__global__ void kexp(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { o[i] = fast_exp(x[i]); o[n + i] = fast_rsqrt(x[i] < 0 ? -x[i] : x[i]); }
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * (size_t)n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = std::exp((u - 0.5) * 40.0);  // 2e-9 .. 5e8
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8);
  hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"rcp raw", "rcp + 1 step", "rcp + 2 steps", "rsq raw", "rsq + 1 step", "rsq + 2 steps"};
  for (int j = 0; j < 6; ++j) {
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      const long double ref = j < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
      const long double err = fabsl(((long double)o[(size_t)j * n + i] - ref) / ref);
      if (err > worst) worst = err;
    }
    printf("%-14s max relative error %.3Le  (= %.2Lf ulp of 2^-53)\n", names[j], worst, worst / 1.1102230246251565e-16L);
  }
  for (int i = 0; i < n; ++i) x[i] = -745.0 + 1455.0 * (double)i / n;
  x[0] = -800.0; x[1] = 800.0; x[2] = 0.0; x[3] = -1e300; x[4] = 1e300; x[5] = 709.78; x[6] = -708.4;
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  kexp<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, 2 * (size_t)n * 8, hipMemcpyDeviceToHost);
  long double worst = 0; int wi = 0;
  for (int i = 7; i < n; ++i) {
    const long double ref = expl((long double)x[i]);
    if (ref < 1e-300L) continue;  // denormal results round coarsely in any implementation
    const long double err = fabsl(((long double)o[i] - ref) / ref);
    if (err > worst) { worst = err; wi = i; }
  }
  printf("fast_exp       max relative error %.3Le (= %.2Lf ulp) at x = %.6f\n", worst, worst / 1.1102230246251565e-16L, x[wi]);
  for (int i = 0; i < 7; ++i) printf("fast_exp(%g) = %g   libm %g\n", x[i], o[i], std::exp(x[i]));
  worst = 0;
  for (int i = 7; i < n; ++i) {
    const long double ax = fabsl((long double)x[i]);
    if (ax == 0) continue;
    const long double ref = 1.0L / sqrtl(ax);
    const long double err = fabsl(((long double)o[n + i] - ref) / ref);
    if (err > worst) worst = err;
  }
  printf("fast_rsqrt     max relative error %.3Le (= %.2Lf ulp)\n", worst, worst / 1.1102230246251565e-16L);
  return 0;
}

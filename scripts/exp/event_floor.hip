// How much of a HIP-event interval around a short kernel is the events' own cost?
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/exp/event_floor.bin scripts/exp/event_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_read(const double2* __restrict__ in, double* out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  double acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = in[i]; acc += v.x + v.y; }
  if (acc == 1.2345) out[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const size_t bytes = 5777760, n = bytes / 16;
  double2* in; double* out;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 8)); CK(hipMemset(in, 0, bytes));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b, c, d; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&d));
  std::vector<float> empty, one, many, tail;
  for (int it = 0; it < 30; ++it) {
    CK(hipEventRecord(a, s)); CK(hipEventRecord(b, s));
    hipLaunchKernelGGL(k_read, dim3(118), dim3(256), 0, s, in, out, n);
    CK(hipEventRecord(c, s));
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_read, dim3(118), dim3(256), 0, s, in, out, n);
    CK(hipEventRecord(d, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b)); empty.push_back(ms * 1e3f);
    CK(hipEventElapsedTime(&ms, b, c)); one.push_back(ms * 1e3f);
    CK(hipEventElapsedTime(&ms, c, d)); many.push_back(ms * 1e3f / 20);
  }
  auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("empty interval %.2f us, one kernel %.2f us, per kernel in a run of 20 %.2f us\n", med(empty), med(one), med(many));
  return 0;
}

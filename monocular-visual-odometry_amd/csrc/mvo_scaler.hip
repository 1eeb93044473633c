// mvo_scaler.hip -- the closed-form scale stage on gfx950: C ABI of include/mvo_hip_scaler.h.
// Replaces the arithmetic of Scaler::stage (/root/reference/mvo/src/pipeline/Scaler.cpp:21-68).
// No CPU fallback: every entry point needs a HIP device.
//
//   k_scaler_small   <= 2048 tracks (the reference's real operating point is <= 60): ONE launch,
//                    one workgroup -- per-track scales, bitonic sort of the order-preserving keys
//                    in LDS, the two middle values
//   k_scaler_tracks  per-track scales, 72 B read + 8 B written per track (HBM-bound)
//   k_scaler_hist / k_scaler_pick   radix select of the middle rank: six 11-bit digits, one pass
//                    over the 8-byte scales each; per-workgroup LDS histograms merged with integer
//                    atomics (order-independent, so the result is deterministic)
//   k_scaler_next    the upper middle value of an even count: number of keys <= the lower one and
//                    the smallest key above it (per-workgroup partials, then one wavefront)
#include "../../include/mvo_hip_scaler.h"
#include "../../include/mvo_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#define SC_SMALL 2048
#define SC_BINS 2048

struct ScalerIn {  // == mvo_hip_scaler_input
  double R0[9], R1[9], R2[9], u1[3], n1, lowest, highest;
};
static_assert(sizeof(ScalerIn) == sizeof(mvo_hip_scaler_input), "input record layout");

struct ScalerSel {
  unsigned long long prefix, mask, rank;  // keys matching (key & mask) == prefix are still candidates
  unsigned long long key1, key2;          // results
  unsigned long long count_le, min_gt;
};

// order-preserving map double -> uint64 (NaN last)
__device__ __forceinline__ unsigned long long sc_key(double x) {
  if (x != x) return ~0ull;
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | (1ull << 63));
}
static double sc_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k ^ (1ull << 63)) : ~k;
  double x;
  std::memcpy(&x, &b, sizeof x);
  return x;
}

// Scaler.cpp:35-43
__device__ __forceinline__ double sc_track(const ScalerIn& in, const double* __restrict__ v0,
                                           const double* __restrict__ v1, const double* __restrict__ v2) {
  double m[3][3];
  const double* R[3] = {in.R0, in.R1, in.R2};
  const double* v[3] = {v0, v1, v2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double x = v[k][0], y = v[k][1], z = v[k][2];
    const double nrm = sqrt(x * x + y * y + z * z);
#pragma unroll
    for (int r = 0; r < 3; ++r) m[k][r] = (R[k][3 * r] * x + R[k][3 * r + 1] * y + R[k][3 * r + 2] * z) / nrm;
  }
  auto dot = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  const double d2u = dot(m[2], in.u1), d10 = dot(m[1], m[0]), d21 = dot(m[2], m[1]), du0 = dot(in.u1, m[0]);
  const double counting = in.n1 * sqrt(1.0 - d2u * d2u) * sqrt(1.0 - d10 * d10);
  const double naming = sqrt(1.0 - d21 * d21) * sqrt(1.0 - du0 * du0);
  return counting / naming;
}

__global__ __launch_bounds__(1024) void k_scaler_small(ScalerIn in, uint32_t n, const double* __restrict__ v0,
                                                      const double* __restrict__ v1,
                                                      const double* __restrict__ v2, double* __restrict__ scales,
                                                      ScalerSel* __restrict__ sel) {
  __shared__ unsigned long long key[SC_SMALL];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < SC_SMALL; i += 1024) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const double s = sc_track(in, v0 + 3 * (size_t)i, v1 + 3 * (size_t)i, v2 + 3 * (size_t)i);
      scales[i] = s;
      k = sc_key(s);
    }
    key[i] = k;
  }
  __syncthreads();
  for (uint32_t size = 2; size <= SC_SMALL; size <<= 1)
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = tid; t < SC_SMALL / 2; t += 1024) {
        const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) {
          key[lo] = b;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  if (tid == 0) {
    sel->key1 = key[(n - 1) / 2];
    sel->key2 = (n % 2 == 0) ? key[(n - 1) / 2 + 1] : key[(n - 1) / 2];
  }
}

__global__ __launch_bounds__(256) void k_scaler_tracks(ScalerIn in, uint64_t n, const double* __restrict__ v0,
                                                      const double* __restrict__ v1,
                                                      const double* __restrict__ v2, double* __restrict__ scales) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    scales[i] = sc_track(in, v0 + 3 * i, v1 + 3 * i, v2 + 3 * i);
}

__global__ __launch_bounds__(256) void k_scaler_hist(uint64_t n, const double* __restrict__ scales,
                                                    const ScalerSel* __restrict__ sel, int shift, int nbits,
                                                    uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[SC_BINS];
  for (int b = threadIdx.x; b < SC_BINS; b += 256) lh[b] = 0;
  __syncthreads();
  const unsigned long long prefix = sel->prefix, mask = sel->mask, dm = (1ull << nbits) - 1;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = sc_key(scales[i]);
    if ((k & mask) == prefix) atomicAdd(&lh[(k >> shift) & dm], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < SC_BINS; b += 256)
    if (lh[b]) atomicAdd(&hist[b], lh[b]);
}

// one wavefront: the bin that holds the wanted rank; narrows the candidate set; clears the histogram
__global__ __launch_bounds__(64) void k_scaler_pick(ScalerSel* __restrict__ sel, int shift, int nbits, int last,
                                                   uint32_t* __restrict__ hist) {
  const int lane = threadIdx.x, nb = 1 << nbits, per = SC_BINS / 64;
  unsigned long long local = 0;
  for (int q = 0; q < per; ++q) {
    const int b = lane * per + q;
    if (b < nb) local += hist[b];
  }
  // exclusive prefix over the 64 lane totals
  unsigned long long incl = local;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const unsigned long long excl = incl - local, rank = sel->rank;
  if (rank >= excl && rank < incl) {  // exactly one lane
    unsigned long long c = excl;
    for (int q = 0; q < per; ++q) {
      const int b = lane * per + q;
      const unsigned long long hcount = b < nb ? hist[b] : 0;
      if (rank < c + hcount) {
        sel->prefix |= (unsigned long long)b << shift;
        sel->mask |= ((1ull << nbits) - 1) << shift;
        sel->rank = rank - c;
        if (last) sel->key1 = sel->prefix;
        break;
      }
      c += hcount;
    }
  }
  __syncthreads();
  for (int b = lane; b < SC_BINS; b += 64) hist[b] = 0;
}

// per workgroup: how many keys are <= the lower middle one, and the smallest key above it
__global__ __launch_bounds__(256) void k_scaler_next(uint64_t n, const double* __restrict__ scales,
                                                    const ScalerSel* __restrict__ sel,
                                                    unsigned long long* __restrict__ part) {
  __shared__ unsigned long long wc[4], wm[4];
  const unsigned long long k1 = sel->key1;
  unsigned long long cnt = 0, mn = ~0ull;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = sc_key(scales[i]);
    if (k <= k1) ++cnt;
    else if (k < mn) mn = k;
  }
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    const unsigned long long t = __shfl_xor(mn, o, 64);
    mn = t < mn ? t : mn;
  }
  if ((threadIdx.x & 63) == 0) {
    wc[threadIdx.x >> 6] = cnt;
    wm[threadIdx.x >> 6] = mn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long c = wc[0] + wc[1] + wc[2] + wc[3], m = wm[0];
    for (int q = 1; q < 4; ++q) m = wm[q] < m ? wm[q] : m;
    part[2 * blockIdx.x] = c;
    part[2 * blockIdx.x + 1] = m;
  }
}

__global__ __launch_bounds__(64) void k_scaler_next_fin(uint32_t blocks, const unsigned long long* __restrict__ part,
                                                       ScalerSel* __restrict__ sel) {
  unsigned long long cnt = 0, mn = ~0ull;
  for (uint32_t g = threadIdx.x; g < blocks; g += 64) {
    cnt += part[2 * g];
    const unsigned long long m = part[2 * g + 1];
    mn = m < mn ? m : mn;
  }
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    const unsigned long long t = __shfl_xor(mn, o, 64);
    mn = t < mn ? t : mn;
  }
  if (threadIdx.x == 0) {
    sel->count_le = cnt;
    sel->min_gt = mn;
  }
}

struct mvo_hip_scaler_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipError_t last_hip_error = hipSuccess;
  double *d_v[3] = {nullptr, nullptr, nullptr}, *d_scales = nullptr;
  uint64_t cap = 0, n = 0;
  uint32_t* d_hist = nullptr;
  unsigned long long* d_part = nullptr;  // 2 per workgroup of k_scaler_next
  ScalerSel* d_sel = nullptr;
  ScalerSel* h_sel = nullptr;  // pinned
  hipEvent_t ev[2] = {};
  double device_seconds = 0.0;
  uint64_t bytes = 0;
  int n_cu = 256;
};

namespace {

#define SC_TRY(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      h->last_hip_error = e_;                                                             \
      return e_ == hipErrorOutOfMemory ? MVO_HIP_ERR_OUT_OF_MEMORY : MVO_HIP_ERR_DEVICE; \
    }                                                                                     \
  } while (0)

int sc_reserve(mvo_hip_scaler_context* h, uint64_t n) {
  if (n <= h->cap) return MVO_HIP_OK;
  for (auto& p : h->d_v) {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  if (h->d_scales) (void)hipFree(h->d_scales);
  h->d_scales = nullptr;
  h->cap = 0;
  const uint64_t want = n + n / 8 + 64;
  for (auto& p : h->d_v) SC_TRY(hipMalloc((void**)&p, sizeof(double) * 3 * want));
  SC_TRY(hipMalloc((void**)&h->d_scales, sizeof(double) * want));
  h->cap = want;
  return MVO_HIP_OK;
}

int sc_upload(mvo_hip_scaler_context* h, uint64_t n, const double* v0, const double* v1, const double* v2) {
  if (n && (!v0 || !v1 || !v2)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (n >= (1ull << 40)) return MVO_HIP_ERR_UNSUPPORTED;
  SC_TRY(hipSetDevice(h->device));
  int rc = sc_reserve(h, n ? n : 1);
  if (rc != MVO_HIP_OK) return rc;
  const double* src[3] = {v0, v1, v2};
  for (int k = 0; k < 3 && n; ++k)
    SC_TRY(hipMemcpyAsync(h->d_v[k], src[k], sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->stream));
  SC_TRY(hipStreamSynchronize(h->stream));
  h->n = n;
  return MVO_HIP_OK;
}

int sc_run(mvo_hip_scaler_context* h, const mvo_hip_scaler_input* in, double* per_track,
           mvo_hip_scaler_result* res) {
  if (!in || !res) return MVO_HIP_ERR_INVALID_ARGUMENT;
  std::memset(res, 0, sizeof *res);
  const uint64_t n = h->n;
  if (n == 0) return MVO_HIP_OK;  // Scaler.cpp:25-26: nothing to do, the scale stays as it is
  SC_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  ScalerIn ki;
  std::memcpy(&ki, in, sizeof ki);
  SC_TRY(hipEventRecord(h->ev[0], s));
  if (n <= SC_SMALL) {
    hipLaunchKernelGGL(k_scaler_small, dim3(1), dim3(1024), 0, s, ki, (uint32_t)n, h->d_v[0], h->d_v[1], h->d_v[2],
                       h->d_scales, h->d_sel);
    SC_TRY(hipGetLastError());
    h->bytes = 80ull * n;
  } else {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, std::min<uint64_t>((uint64_t)h->n_cu * 16, 8192));
    ScalerSel init{};
    init.rank = (n - 1) / 2;
    init.min_gt = ~0ull;
    *h->h_sel = init;
    SC_TRY(hipMemcpyAsync(h->d_sel, h->h_sel, sizeof(ScalerSel), hipMemcpyHostToDevice, s));
    SC_TRY(hipMemsetAsync(h->d_hist, 0, sizeof(uint32_t) * SC_BINS, s));
    hipLaunchKernelGGL(k_scaler_tracks, dim3(blocks), dim3(256), 0, s, ki, n, h->d_v[0], h->d_v[1], h->d_v[2],
                       h->d_scales);
    SC_TRY(hipGetLastError());
    const int shifts[6] = {53, 42, 31, 20, 9, 0}, bits[6] = {11, 11, 11, 11, 11, 9};
    for (int p = 0; p < 6; ++p) {
      hipLaunchKernelGGL(k_scaler_hist, dim3(blocks), dim3(256), 0, s, n, h->d_scales, h->d_sel, shifts[p], bits[p],
                         h->d_hist);
      hipLaunchKernelGGL(k_scaler_pick, dim3(1), dim3(64), 0, s, h->d_sel, shifts[p], bits[p], p == 5 ? 1 : 0,
                         h->d_hist);
    }
    SC_TRY(hipGetLastError());
    if (n % 2 == 0) {
      hipLaunchKernelGGL(k_scaler_next, dim3(blocks), dim3(256), 0, s, n, h->d_scales, h->d_sel, h->d_part);
      hipLaunchKernelGGL(k_scaler_next_fin, dim3(1), dim3(64), 0, s, blocks, h->d_part, h->d_sel);
      SC_TRY(hipGetLastError());
    }
    h->bytes = (72ull + 8 + 6 * 8 + (n % 2 == 0 ? 8 : 0)) * n;
  }
  SC_TRY(hipEventRecord(h->ev[1], s));
  SC_TRY(hipMemcpyAsync(h->h_sel, h->d_sel, sizeof(ScalerSel), hipMemcpyDeviceToHost, s));
  if (per_track) SC_TRY(hipMemcpyAsync(per_track, h->d_scales, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  SC_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  SC_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->device_seconds = ms * 1e-3;
  unsigned long long k1 = h->h_sel->key1, k2 = h->h_sel->key2;
  if (n > SC_SMALL) k2 = (n % 2 == 0) ? (h->h_sel->count_le >= (n - 1) / 2 + 2 ? k1 : h->h_sel->min_gt) : k1;
  // Scaler.cpp:49-67
  const double a = sc_unkey(k1), b = sc_unkey(k2);
  const double n0 = (n % 2 == 0) ? (a + b) / 2 : a;
  res->median = n0;
  res->n_tracks = n;
  if (n0 > in->highest_length) {
    res->scale = in->highest_length;
    res->clamped = 1;
  } else if (n0 < in->lowest_length) {
    res->scale = in->lowest_length;
    res->clamped = -1;
  } else {
    res->scale = n0;
  }
  return MVO_HIP_OK;
}

}  // namespace

extern "C" {

int mvo_hip_scaler_create(int device_ordinal, mvo_hip_scaler_handle* out) {
  if (!out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MVO_HIP_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= count) return MVO_HIP_ERR_INVALID_ARGUMENT;
  mvo_hip_scaler_context* h = new (std::nothrow) mvo_hip_scaler_context();
  if (!h) return MVO_HIP_ERR_OUT_OF_MEMORY;
  h->device = device_ordinal;
  bool ok = hipSetDevice(device_ordinal) == hipSuccess &&
            hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc((void**)&h->d_hist, sizeof(uint32_t) * SC_BINS) == hipSuccess &&
            hipMalloc((void**)&h->d_sel, sizeof(ScalerSel)) == hipSuccess &&
            hipMalloc((void**)&h->d_part, sizeof(unsigned long long) * 2 * 8192) == hipSuccess &&
            hipHostMalloc((void**)&h->h_sel, sizeof(ScalerSel), hipHostMallocDefault) == hipSuccess &&
            hipEventCreate(&h->ev[0]) == hipSuccess && hipEventCreate(&h->ev[1]) == hipSuccess;
  if (!ok) {
    mvo_hip_scaler_destroy(h);
    return MVO_HIP_ERR_DEVICE;
  }
  (void)hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device_ordinal);
  *out = h;
  return MVO_HIP_OK;
}

void mvo_hip_scaler_destroy(mvo_hip_scaler_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& p : h->d_v)
    if (p) (void)hipFree(p);
  if (h->d_scales) (void)hipFree(h->d_scales);
  if (h->d_hist) (void)hipFree(h->d_hist);
  if (h->d_sel) (void)hipFree(h->d_sel);
  if (h->d_part) (void)hipFree(h->d_part);
  if (h->h_sel) (void)hipHostFree(h->h_sel);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int mvo_hip_scaler_upload(mvo_hip_scaler_handle h, uint64_t n_tracks, const double* v0, const double* v1,
                          const double* v2) {
  return h ? sc_upload(h, n_tracks, v0, v1, v2) : MVO_HIP_ERR_INVALID_ARGUMENT;
}

int mvo_hip_scaler_run(mvo_hip_scaler_handle h, const mvo_hip_scaler_input* in, double* per_track,
                       mvo_hip_scaler_result* result) {
  return h ? sc_run(h, in, per_track, result) : MVO_HIP_ERR_INVALID_ARGUMENT;
}

int mvo_hip_scaler_scale(mvo_hip_scaler_handle h, const mvo_hip_scaler_input* in, uint64_t n_tracks,
                         const double* v0, const double* v1, const double* v2, double* per_track,
                         mvo_hip_scaler_result* result) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = sc_upload(h, n_tracks, v0, v1, v2);
  if (rc != MVO_HIP_OK) return rc;
  return sc_run(h, in, per_track, result);
}

int mvo_hip_scaler_last_timing(mvo_hip_scaler_handle h, double* device_seconds, uint64_t* bytes) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (device_seconds) *device_seconds = h->device_seconds;
  if (bytes) *bytes = h->bytes;
  return MVO_HIP_OK;
}

}  // extern "C"

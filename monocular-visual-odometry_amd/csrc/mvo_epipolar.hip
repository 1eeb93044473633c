// mvo_epipolar.hip -- RANSAC baseline-direction stage on gfx950: C ABI of include/mvo_hip_epipolar.h.
// Replaces EpipolarGeometry::estimateBaseLine / ::calculateBaseLine
// (/root/reference/mvo/src/algorithms/EpipolarGeometry.cpp:34-176).  No CPU fallback.
//
//   k_epi_hypotheses  one thread per draw: A = sum of the three rows a a^T, det test, null
//                     direction by Jacobi rotations
//   k_epi_score_tiled workgroup per (2048 correspondences, 8 hypotheses): every loaded correspondence
//                     is scored against 8 hypotheses held in scalar registers (pInlier, count and summed
//                     probability), partial sums per tile; k_epi_score_reduce adds the tiles in order
//   k_epi_final       the best direction again: inlier flags and per-workgroup partial sums of
//                     a a^T over the inliers (calculateBaseLine over the inlier set)
// The adaptive-iteration bookkeeping of the reference's loop is replayed on the host over the
// per-hypothesis scores (a few hundred integer/double compares), as is the final 3 x 3 null
// direction.
#include "../../include/mvo_hip_epipolar.h"
#include "../../include/mvo_hip.h"

#include <hip/hip_runtime.h>

#include "mvo_alloc.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "mvo_eig3.h"
#include "mvo_wave.h"

__global__ __launch_bounds__(256) void k_epi_hypotheses(uint32_t K, const double* __restrict__ mhi,
                                                       const double* __restrict__ mt,
                                                       const uint32_t* __restrict__ samples,
                                                       double* __restrict__ hyp_b) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const uint32_t i = samples[3 * (size_t)k + q];
    double a[3];
    mvo_epi_row(mt + 3 * (size_t)i, mhi + 3 * (size_t)i, a);  // calculateBaseLine(mtSet, mhiSet): m1 = mt, m2 = mhi
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) A[3 * r + c] += a[r] * a[c];
  }
  double b[3] = {0.0, 0.0, 0.0};
  if (mvo_det3(A) != 0) mvo_eig3_smallest(A, b);
  hyp_b[3 * (size_t)k] = b[0];
  hyp_b[3 * (size_t)k + 1] = b[1];
  hyp_b[3 * (size_t)k + 2] = b[2];
}

#define EP_HB 8   // hypotheses per workgroup: every loaded correspondence is scored against all of them;
                  // their 24 doubles stay in scalar registers for the whole kernel
#define EP_CPT 8  // correspondences per thread

// 1 / sqrt(a) for a normal, positive a: hardware estimate + two Newton steps (full double precision
// to a couple of ulps; the reference's own two square roots and division round no better)
__device__ __forceinline__ double ep_rsqrt(double a) {
  double y = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * a;
  y = y * fma(-h * y, y, 1.5);
#ifndef MVO_EPI_ONE_NEWTON
  y = y * fma(-h * y, y, 1.5);
#endif
  return y;
}

// Workgroup (tile, group): 256 * EP_CPT correspondences x EP_HB hypotheses.  The three dot products
// that do not depend on the hypothesis (m1.m2, |m1|^2, |m2|^2 of EpipolarGeometry.cpp:90-94) are
// formed once per correspondence; per hypothesis two dot products, three fused multiply-adds and one
// reciprocal square root of the PRODUCT of the two radicands remain.  Radicands that are not
// comfortably positive (a ray along the baseline) take the reference's expression literally, NaN
// and infinity included.  Partial sums per (tile, hypothesis); k_epi_score_reduce adds the tiles in
// order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void k_epi_score_tiled(uint32_t N, uint32_t K, const double* __restrict__ mhi,
                                                        const double* __restrict__ mt,
                                                        const double* __restrict__ hyp_b, double threshold,
                                                        double* __restrict__ part_prob,
                                                        uint32_t* __restrict__ part_cnt) {
  __shared__ double wp[4][EP_HB];
  __shared__ uint32_t wc[4][EP_HB];
  const uint32_t tile = blockIdx.x, k0 = blockIdx.y * EP_HB;
  double prob[EP_HB];
  uint32_t cnt[EP_HB];
#pragma unroll
  for (int h = 0; h < EP_HB; ++h) {
    prob[h] = 0.0;
    cnt[h] = 0;
  }
  double hb[EP_HB][3];  // wavefront-uniform
#pragma unroll
  for (int h = 0; h < EP_HB; ++h) {
    const uint32_t k = min(k0 + h, K - 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) hb[h][j] = hyp_b[3 * (size_t)k + j];
  }
  for (uint32_t c = 0; c < EP_CPT; ++c) {
    const uint32_t i = (tile * EP_CPT + c) * 256 + threadIdx.x;
    if (i >= N) break;
    const double* __restrict__ m1 = mhi + 3 * (size_t)i;
    const double* __restrict__ m2 = mt + 3 * (size_t)i;
    const double a0 = m1[0], a1 = m1[1], a2 = m1[2], c0 = m2[0], c1 = m2[1], c2 = m2[2];
    const double m12 = a0 * c0 + a1 * c1 + a2 * c2;
    const double n1 = a0 * a0 + a1 * a1 + a2 * a2;
    const double n2 = c0 * c0 + c1 * c1 + c2 * c2;
    uint32_t odd = 0;  // hypotheses whose radicands are not comfortably positive for this correspondence
#pragma unroll
    for (int h = 0; h < EP_HB; ++h) {  // branch-free: sixteen independent chains for the scheduler
      const double b0 = hb[h][0], b1 = hb[h][1], b2 = hb[h][2];
      const double m1b = a0 * b0 + a1 * b1 + a2 * b2;
      const double m2b = c0 * b0 + c1 * b1 + c2 * b2;
      const double x = fma(-m1b, m1b, n1), y = fma(-m2b, m2b, n2), num = fma(-m2b, m1b, m12);
      const double p = num * ep_rsqrt(x * y);
      // comfortably positive radicands, finite product -- and p not within 1e-12 of the threshold:
      // `p > threshold` feeds an INTEGER (the inlier count, hence the adaptive iteration count and the
      // winner, EpipolarGeometry.cpp:100-118), so a correspondence that close takes the reference's own
      // expression below, the one k_epi_final and the oracle decide with
      const bool ok = x > 1e-140 && y > 1e-140 && x * y < 1e280 &&
                      fabs(p - threshold) > 1e-12 * fmax(1.0, fabs(threshold));
      const bool in = ok && p > threshold;
      cnt[h] += in ? 1u : 0u;
      prob[h] += in ? p : 0.0;
      odd |= ok ? 0u : (1u << h);
    }
    if (odd) {  // rare: the reference's expression literally, NaN and infinity included
#pragma unroll
      for (int h = 0; h < EP_HB; ++h)
        if ((odd >> h) & 1u) {
          const double p = mvo_epi_pinlier(m1, m2, hb[h]);
          if (p > threshold) {
            ++cnt[h];
            prob[h] += p;
          }
        }
    }
  }
  {
    // 8 sums and 8 counts over the wavefront as two transposing register butterflies (mvo_wave.h) instead
    // of 8 x 2 shuffle butterflies through the LDS crossbar; the counts ride as doubles (exact: < 2^53)
    static_assert(EP_HB == 8, "wave_reduce8 takes the workgroup's eight hypotheses at once");
    double cd[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) cd[h] = (double)cnt[h];
    const int lane = (int)(threadIdx.x & 63);
    const double tp = wave_reduce8(prob, lane), tc = wave_reduce8(cd, lane);
    if ((lane & 7) == 0) {
      wp[threadIdx.x >> 6][reduce8_value_of_lane(lane)] = tp;
      wc[threadIdx.x >> 6][reduce8_value_of_lane(lane)] = (uint32_t)tc;
    }
  }
  __syncthreads();
  if (threadIdx.x < EP_HB && k0 + threadIdx.x < K) {
    const uint32_t h = threadIdx.x;
    part_prob[(size_t)tile * K + k0 + h] = (wp[0][h] + wp[1][h]) + (wp[2][h] + wp[3][h]);
    part_cnt[(size_t)tile * K + k0 + h] = wc[0][h] + wc[1][h] + wc[2][h] + wc[3][h];
  }
}

__global__ __launch_bounds__(256) void k_epi_score_reduce(uint32_t K, uint32_t tiles,
                                                         const double* __restrict__ part_prob,
                                                         const uint32_t* __restrict__ part_cnt,
                                                         uint32_t* __restrict__ hyp_inl,
                                                         double* __restrict__ hyp_prob) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double prob = 0.0;
  uint32_t cnt = 0;
#pragma unroll 8
  for (uint32_t t = 0; t < tiles; ++t) {  // loads of eight tiles in flight; the sum itself stays in tile order
    prob += part_prob[(size_t)t * K + k];
    cnt += part_cnt[(size_t)t * K + k];
  }
  hyp_prob[k] = prob;
  hyp_inl[k] = cnt;
}

// inlier flags of direction b and per-workgroup sums of the 6 distinct entries of a a^T over them
__global__ __launch_bounds__(256) void k_epi_final(uint32_t N, const double* __restrict__ mhi,
                                                  const double* __restrict__ mt, double b0, double b1, double b2,
                                                  double threshold, uint8_t* __restrict__ flags,
                                                  double* __restrict__ part) {
  __shared__ double ws[4][6];
  const double b[3] = {b0, b1, b2};
  double s[6] = {0, 0, 0, 0, 0, 0};
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < N; i += stride) {
    const double p = mvo_epi_pinlier(mhi + 3 * (size_t)i, mt + 3 * (size_t)i, b);
    const bool in = p > threshold;
    flags[i] = in ? 1 : 0;
    if (in) {
      double a[3];
      mvo_epi_row(mt + 3 * (size_t)i, mhi + 3 * (size_t)i, a);
      s[0] += a[0] * a[0];
      s[1] += a[0] * a[1];
      s[2] += a[0] * a[2];
      s[3] += a[1] * a[1];
      s[4] += a[1] * a[2];
      s[5] += a[2] * a[2];
    }
  }
#pragma unroll
  for (int v = 0; v < 6; ++v)
    for (int o = 32; o > 0; o >>= 1) s[v] += __shfl_xor(s[v], o, 64);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int v = 0; v < 6; ++v) ws[threadIdx.x >> 6][v] = s[v];
  __syncthreads();
  if (threadIdx.x < 6) part[6 * (size_t)blockIdx.x + threadIdx.x] =
      (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
}

struct mvo_hip_epipolar_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipError_t last_hip_error = hipSuccess;
  double *d_mhi = nullptr, *d_mt = nullptr, *d_hb = nullptr, *d_hp = nullptr, *d_part = nullptr;
  uint32_t *d_samples = nullptr, *d_hi = nullptr, *d_part_cnt = nullptr;
  double* d_part_prob = nullptr;
  size_t cap_part = 0;
  uint8_t* d_flags = nullptr;
  size_t cap_n = 0, cap_k = 0;
  hipEvent_t ev[2] = {};
  double device_seconds = 0.0;
  uint64_t evaluations = 0;
};

namespace {

#define EP_TRY(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      h->last_hip_error = e_;                                                             \
      return e_ == hipErrorOutOfMemory ? MVO_HIP_ERR_OUT_OF_MEMORY : MVO_HIP_ERR_DEVICE; \
    }                                                                                     \
  } while (0)

constexpr uint32_t kFinalBlocks = 256;

template <class T>
hipError_t regrow(T*& p, size_t n) {
  if (p) (void)mvo_device_free(p);
  p = nullptr;
  return mvo_device_alloc((void**)&p, sizeof(T) * n);
}

unsigned int binom3(unsigned int n) {  // boost::math::binomial_coefficient<double>(n, 3) -> unsigned int
  if (n < 3) return 0;
  const double v = (double)n * (double)(n - 1) * (double)(n - 2) / 6.0;
  return (unsigned int)(unsigned long long)v;
}

unsigned int iterations_for(unsigned int N, unsigned int nInlier, double ps) {  // EpipolarGeometry.cpp:11-32
  const double p = (double)binom3(nInlier) / (double)binom3(N);
  const double it = std::ceil(std::log(1 - ps) / std::log(1 - p));
  if (!(it >= 0.0)) return 0;
  if (it >= 4294967295.0) return 4294967295u;
  return (unsigned int)it;
}

int ep_score(mvo_hip_epipolar_context* h, uint32_t n, const double* mhi, const double* mt, uint32_t K,
             const uint32_t* samples, double threshold) {
  if (!mhi || !mt || (K && !samples)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  for (size_t q = 0; q < 3 * (size_t)K; ++q)
    if (samples[q] >= n) return MVO_HIP_ERR_INVALID_ARGUMENT;
  EP_TRY(hipSetDevice(h->device));
  if (n > h->cap_n) {
    const size_t want = n + n / 8 + 64;
    EP_TRY(regrow(h->d_mhi, 3 * want));
    EP_TRY(regrow(h->d_mt, 3 * want));
    EP_TRY(regrow(h->d_flags, want));
    h->cap_n = want;
  }
  if (K > h->cap_k) {
    const size_t want = K + K / 8 + 64;
    EP_TRY(regrow(h->d_samples, 3 * want));
    EP_TRY(regrow(h->d_hb, 3 * want));
    EP_TRY(regrow(h->d_hp, want));
    EP_TRY(regrow(h->d_hi, want));
    h->cap_k = want;
  }
  const uint32_t tiles = (n + 256 * EP_CPT - 1) / (256 * EP_CPT);
  if ((size_t)tiles * K > h->cap_part) {
    const size_t want = (size_t)tiles * K + 1024;
    EP_TRY(regrow(h->d_part_prob, want));
    EP_TRY(regrow(h->d_part_cnt, want));
    h->cap_part = want;
  }
  hipStream_t s = h->stream;
  EP_TRY(hipMemcpyAsync(h->d_mhi, mhi, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
  EP_TRY(hipMemcpyAsync(h->d_mt, mt, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
  if (K) EP_TRY(hipMemcpyAsync(h->d_samples, samples, sizeof(uint32_t) * 3 * K, hipMemcpyHostToDevice, s));
  EP_TRY(hipEventRecord(h->ev[0], s));
  if (K) {
    hipLaunchKernelGGL(k_epi_hypotheses, dim3((K + 255) / 256), dim3(256), 0, s, K, h->d_mhi, h->d_mt, h->d_samples,
                       h->d_hb);
    hipLaunchKernelGGL(k_epi_score_tiled, dim3(tiles, (K + EP_HB - 1) / EP_HB), dim3(256), 0, s, n, K, h->d_mhi, h->d_mt,
                       h->d_hb, threshold, h->d_part_prob, h->d_part_cnt);
    hipLaunchKernelGGL(k_epi_score_reduce, dim3((K + 255) / 256), dim3(256), 0, s, K, tiles, h->d_part_prob,
                       h->d_part_cnt, h->d_hi, h->d_hp);
    EP_TRY(hipGetLastError());
  }
  EP_TRY(hipEventRecord(h->ev[1], s));
  h->evaluations = (uint64_t)K * n;
  return MVO_HIP_OK;
}

}  // namespace

extern "C" {

int mvo_hip_epipolar_create(int device_ordinal, mvo_hip_epipolar_handle* out) {
  if (!out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MVO_HIP_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= count) return MVO_HIP_ERR_INVALID_ARGUMENT;
  mvo_hip_epipolar_context* h = new (std::nothrow) mvo_hip_epipolar_context();
  if (!h) return MVO_HIP_ERR_OUT_OF_MEMORY;
  h->device = device_ordinal;
  const bool ok = hipSetDevice(device_ordinal) == hipSuccess &&
                  hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
                  mvo_device_alloc((void**)&h->d_part, sizeof(double) * 6 * kFinalBlocks) == hipSuccess &&
                  hipEventCreate(&h->ev[0]) == hipSuccess && hipEventCreate(&h->ev[1]) == hipSuccess;
  if (!ok) {
    mvo_hip_epipolar_destroy(h);
    return MVO_HIP_ERR_DEVICE;
  }
  *out = h;
  return MVO_HIP_OK;
}

void mvo_hip_epipolar_destroy(mvo_hip_epipolar_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* bufs[] = {h->d_mhi, h->d_mt, h->d_hb, h->d_hp, h->d_part, h->d_samples, h->d_hi, h->d_flags,
                  h->d_part_prob, h->d_part_cnt};
  for (void* p : bufs)
    if (p) (void)mvo_device_free(p);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int mvo_hip_epipolar_score(mvo_hip_epipolar_handle h, uint32_t n, const double* mhi, const double* mt,
                           uint32_t n_samples, const uint32_t* samples, double threshold, double* hyp_b,
                           uint32_t* hyp_inliers, double* hyp_probability) {
  if (!h || n == 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = ep_score(h, n, mhi, mt, n_samples, samples, threshold);
  if (rc != MVO_HIP_OK) return rc;
  hipStream_t s = h->stream;
  if (n_samples) {
    if (hyp_b) EP_TRY(hipMemcpyAsync(hyp_b, h->d_hb, sizeof(double) * 3 * n_samples, hipMemcpyDeviceToHost, s));
    if (hyp_inliers) EP_TRY(hipMemcpyAsync(hyp_inliers, h->d_hi, sizeof(uint32_t) * n_samples, hipMemcpyDeviceToHost, s));
    if (hyp_probability)
      EP_TRY(hipMemcpyAsync(hyp_probability, h->d_hp, sizeof(double) * n_samples, hipMemcpyDeviceToHost, s));
  }
  EP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  EP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->device_seconds = ms * 1e-3;
  return MVO_HIP_OK;
}

int mvo_hip_epipolar_estimate(mvo_hip_epipolar_handle h, uint32_t n, const double* mhi, const double* mt,
                              uint32_t n_samples, const uint32_t* samples, double ps, double threshold,
                              uint32_t* inlier_indexes, mvo_hip_epipolar_result* result) {
  if (!h || !result) return MVO_HIP_ERR_INVALID_ARGUMENT;
  std::memset(result, 0, sizeof *result);
  result->best_probability = -1;
  if (n < 3) return MVO_HIP_OK;  // EpipolarGeometry.cpp:44-48: too few features, b = (0, 0, 0)
  if (!inlier_indexes) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = ep_score(h, n, mhi, mt, n_samples, samples, threshold);
  if (rc != MVO_HIP_OK) return rc;
  hipStream_t s = h->stream;
  std::vector<uint32_t> hi(n_samples);
  std::vector<double> hp(n_samples), hb(3 * (size_t)n_samples);
  if (n_samples) {
    EP_TRY(hipMemcpyAsync(hi.data(), h->d_hi, sizeof(uint32_t) * n_samples, hipMemcpyDeviceToHost, s));
    EP_TRY(hipMemcpyAsync(hp.data(), h->d_hp, sizeof(double) * n_samples, hipMemcpyDeviceToHost, s));
    EP_TRY(hipMemcpyAsync(hb.data(), h->d_hb, sizeof(double) * 3 * n_samples, hipMemcpyDeviceToHost, s));
  }
  EP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  EP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->device_seconds = ms * 1e-3;
  // replay of the reference's loop bookkeeping over the scores, in draw order (:41-118)
  double inlierProbability = 0.3;
  if (inlierProbability * n < 3) inlierProbability = 1;
  unsigned int nIterations = iterations_for(n, (unsigned int)(int)(n * inlierProbability), ps);
  unsigned int iteration = 0;
  double pBest = -1;
  long best = -1;
  while (iteration < nIterations) {
    if (iteration >= n_samples) {
      result->exhausted = 1;
      break;
    }
    const unsigned int nInlier = hi[iteration];
    const double nProbability = hp[iteration];
    iteration++;
    if ((double)nInlier / (double)n > inlierProbability) {
      inlierProbability = (double)nInlier / (double)n;
      nIterations = iterations_for(n, nInlier, ps);
    }
    if (nProbability > pBest) {
      pBest = nProbability;
      best = (long)iteration - 1;
    }
  }
  result->iterations = iteration;
  result->planned_iterations = nIterations;
  result->best_probability = pBest;
  if (best < 0) return MVO_HIP_OK;  // no hypothesis consumed: empty inlier set, b = 0 (:122-133 on empty sets)
  result->best_hypothesis = (uint32_t)best;
  // inliers of the winner and calculateBaseLine over them (:122-133)
  const uint32_t blocks = std::min<uint32_t>(kFinalBlocks, (n + 255) / 256);
  hipLaunchKernelGGL(k_epi_final, dim3(blocks), dim3(256), 0, s, n, h->d_mhi, h->d_mt, hb[3 * best], hb[3 * best + 1],
                     hb[3 * best + 2], threshold, h->d_flags, h->d_part);
  EP_TRY(hipGetLastError());
  std::vector<uint8_t> flags(n);
  std::vector<double> part(6 * (size_t)blocks);
  EP_TRY(hipMemcpyAsync(flags.data(), h->d_flags, n, hipMemcpyDeviceToHost, s));
  EP_TRY(hipMemcpyAsync(part.data(), h->d_part, sizeof(double) * 6 * blocks, hipMemcpyDeviceToHost, s));
  EP_TRY(hipStreamSynchronize(s));
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (flags[i]) inlier_indexes[m++] = i;
  result->n_inliers = m;
  double S[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t g = 0; g < blocks; ++g)
    for (int v = 0; v < 6; ++v) S[v] += part[6 * (size_t)g + v];
  const double A[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
  if (mvo_det3(A) != 0) mvo_eig3_smallest(A, result->b);
  return MVO_HIP_OK;
}

int mvo_hip_epipolar_last_timing(mvo_hip_epipolar_handle h, double* device_seconds, uint64_t* evaluations) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (device_seconds) *device_seconds = h->device_seconds;
  if (evaluations) *evaluations = h->evaluations;
  return MVO_HIP_OK;
}

}  // extern "C"

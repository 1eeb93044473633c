// mvo_tracks.hip -- device-resident feature-track tables and the on-device observation gather:
// C ABI of include/mvo_hip_tracks.h.  Replaces the host-side chain walk of
// IterativeRefinement::refine (/root/reference/mvo/src/algorithms/IterativeRefinement.cpp:116-165,
// Frame::getCorrespondingFeatures sliding_window/Frame.hpp:258-302, normFeatures
// operations/FeatureOperations.cpp:117-122).  No CPU fallback.
//
//   k_tracks_counter  on push: _preFeatureCounter of the new frame (Frame.cpp:132-148)
//   k_tracks_count    one workgroup per frame pair (a, b): how many features of frame a reach back
//                     b - a + 1 frames (Frame.hpp:277-279)
//   k_tracks_fill     one workgroup per frame pair: ordered compaction (feature order of frame a,
//                     as the reference's loop pushes them), chain walk to frame b + 1
//                     (Frame.hpp:283-289), both rays normalised, written at the pair's offset
// All kernels run on the solver's stream; the solve is queued right behind the gather.
#include "../../include/mvo_hip_tracks.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "mvo_internal.h"

#define TR_THREADS 256

struct TracksDev {
  const double* rays;     // [max_frames][max_features][3]
  const int32_t* pre;     // [max_frames][max_features]
  const uint32_t* counter;
  const uint32_t* n_feat; // [max_frames]
  uint32_t max_frames, max_features, head;  // slot of the newest frame
};

__device__ __forceinline__ uint32_t tr_slot(const TracksDev& t, uint32_t age) {  // age 0 = newest
  return (t.head + t.max_frames - age) % t.max_frames;
}

// FeatureOperations.cpp:119: scale = 1.0 / cv::norm(v).  Contraction is switched off here: a fused
// multiply-add in the sum of squares rounds differently from the host's separate multiply and add
// (the _rn intrinsics are plain operators to the compiler and would be fused all the same).
__device__ __forceinline__ double tr_inv_norm(const double* __restrict__ v) {
#pragma clang fp contract(off)
  const double xx = v[0] * v[0], yy = v[1] * v[1], zz = v[2] * v[2];
  const double ss = (xx + yy) + zz;
  return 1.0 / __builtin_sqrt(ss);
}

// (a, b) of pair p in the reference's loop order: for a < M: for b in [a, N-1)
__device__ __forceinline__ void tr_pair(uint32_t M, uint32_t N, uint32_t p, uint32_t& a, uint32_t& b) {
  a = 0;
  uint32_t start = 0;
  for (uint32_t k = 0; k < M; ++k) {
    const uint32_t len = N - 1 - k;
    if (p >= start + len) start += len;
    else {
      a = k;
      break;
    }
  }
  b = a + (p - start);
}

__global__ __launch_bounds__(TR_THREADS) void k_tracks_counter(uint32_t n, const int32_t* __restrict__ pre,
                                                              const uint32_t* __restrict__ prev_counter,
                                                              uint32_t prev_n, int32_t* __restrict__ pre_out,
                                                              uint32_t* __restrict__ counter_out) {
  const uint32_t i = blockIdx.x * TR_THREADS + threadIdx.x;
  if (i >= n) return;
  const int32_t p = pre[i];
  if (p >= 0 && (uint32_t)p < prev_n) {  // Frame.cpp:138-139
    counter_out[i] = prev_counter[p] + 1;
    pre_out[i] = p;
  } else {  // Frame.cpp:141-144
    counter_out[i] = 0;
    pre_out[i] = -1;
  }
}

__global__ __launch_bounds__(TR_THREADS) void k_tracks_count(TracksDev t, uint32_t M, uint32_t N,
                                                            uint32_t* __restrict__ counts) {
  __shared__ uint32_t wsum[TR_THREADS / 64];
  uint32_t a, b;
  tr_pair(M, N, blockIdx.x, a, b);
  const uint32_t slot = tr_slot(t, a), n = t.n_feat[slot], past = b - a + 1;
  const uint32_t* __restrict__ cnt = t.counter + (size_t)slot * t.max_features;
  uint32_t c = 0;
  for (uint32_t i = threadIdx.x; i < n; i += TR_THREADS) c += cnt[i] >= past ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(TR_THREADS) void k_tracks_fill(TracksDev t, uint32_t M, uint32_t N,
                                                           const uint64_t* __restrict__ offsets,
                                                           double* __restrict__ m0, double* __restrict__ m1) {
  __shared__ uint32_t wsum[TR_THREADS / 64];
  __shared__ uint32_t base_sh;
  uint32_t a, b;
  tr_pair(M, N, blockIdx.x, a, b);
  const uint32_t slot_a = tr_slot(t, a), n = t.n_feat[slot_a], past = b - a + 1;
  const uint32_t* __restrict__ cnt = t.counter + (size_t)slot_a * t.max_features;
  const uint64_t off = offsets[blockIdx.x];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_sh = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n; i0 += TR_THREADS) {
    const uint32_t i = i0 + threadIdx.x;
    const bool take = i < n && cnt[i] >= past;
    // rank of this feature among the taken ones of the chunk, in feature order
    const unsigned long long ballot = __ballot(take);
    const uint32_t in_wave = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(ballot);
    __syncthreads();
    uint32_t before = base_sh;
    for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
    if (take) {
      uint32_t idx = i;  // Frame.hpp:283-289: follow _preFeature `past` frames back
      for (uint32_t s = 0; s < past; ++s) idx = (uint32_t)t.pre[(size_t)tr_slot(t, a + s) * t.max_features + idx];
      const double* __restrict__ r0 = t.rays + ((size_t)slot_a * t.max_features + i) * 3;
      const double* __restrict__ r1 = t.rays + ((size_t)tr_slot(t, b + 1) * t.max_features + idx) * 3;
      const double s0 = tr_inv_norm(r0), s1 = tr_inv_norm(r1);
      const size_t o = 3 * (size_t)(off + before + in_wave);
      m0[o] = r0[0] * s0;
      m0[o + 1] = r0[1] * s0;
      m0[o + 2] = r0[2] * s0;
      m1[o] = r1[0] * s1;
      m1[o + 1] = r1[1] * s1;
      m1[o + 2] = r1[2] * s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_sh += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

struct mvo_hip_tracks_context {
  mvo_hip_handle solver = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipError_t last_hip_error = hipSuccess;
  uint32_t max_frames = 0, max_features = 0, head = 0, n_pushed = 0;
  double* d_rays = nullptr;
  int32_t* d_pre = nullptr;
  uint32_t *d_counter = nullptr, *d_nfeat = nullptr, *d_counts = nullptr;
  int32_t* d_pre_in = nullptr;  // staging of the caller's predecessor indices
  uint64_t* d_offsets = nullptr;
  double *d_m0 = nullptr, *d_m1 = nullptr;
  uint64_t cap_obs = 0;
  uint32_t cap_pairs = 0;
  std::vector<uint32_t> h_nfeat;
  uint32_t* h_counts = nullptr;  // pinned
  uint64_t* h_offsets = nullptr; // pinned
  bool profiling = false, events_ready = false, count_timed = false, fill_timed = false;
  hipEvent_t ev[4] = {};
  uint64_t last_blocks = 0, last_fill_bytes = 0;
};

namespace {

#define TR_TRY(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      t->last_hip_error = e_;                                                             \
      return e_ == hipErrorOutOfMemory ? MVO_HIP_ERR_OUT_OF_MEMORY : MVO_HIP_ERR_DEVICE; \
    }                                                                                     \
  } while (0)

uint32_t num_pairs(uint32_t M, uint32_t N) { return M * (N - 1) - (M * (M - 1)) / 2; }

TracksDev dev_view(const mvo_hip_tracks_context* t) {
  TracksDev d;
  d.rays = t->d_rays;
  d.pre = t->d_pre;
  d.counter = t->d_counter;
  d.n_feat = t->d_nfeat;
  d.max_frames = t->max_frames;
  d.max_features = t->max_features;
  d.head = t->head;
  return d;
}

// k_tracks_count -> host prefix sum: h_offsets / d_offsets hold the CSR offsets of the P pairs
int count_pairs(mvo_hip_tracks_context* t, uint32_t M, uint32_t N) {
  if (M == 0 || N <= M) return MVO_HIP_ERR_INVALID_ARGUMENT;            // IterativeRefinement.cpp:29-30
  if (M > MVO_HIP_MAX_REFINE || N > MVO_HIP_MAX_FRAMES) return MVO_HIP_ERR_UNSUPPORTED;
  if (N > t->n_pushed || N > t->max_frames) return MVO_HIP_ERR_STATE;  // not enough keyframes in the table
  TR_TRY(hipSetDevice(t->device));
  const uint32_t P = num_pairs(M, N);
  if (P > t->cap_pairs) {
    if (t->d_counts) (void)hipFree(t->d_counts);
    if (t->d_offsets) (void)hipFree(t->d_offsets);
    if (t->h_counts) (void)hipHostFree(t->h_counts);
    if (t->h_offsets) (void)hipHostFree(t->h_offsets);
    t->d_counts = nullptr;
    t->d_offsets = nullptr;
    t->h_counts = nullptr;
    t->h_offsets = nullptr;
    t->cap_pairs = 0;
    TR_TRY(hipMalloc((void**)&t->d_counts, sizeof(uint32_t) * (P + 1)));
    TR_TRY(hipMalloc((void**)&t->d_offsets, sizeof(uint64_t) * (P + 1)));
    TR_TRY(hipHostMalloc((void**)&t->h_counts, sizeof(uint32_t) * (P + 1), hipHostMallocDefault));
    TR_TRY(hipHostMalloc((void**)&t->h_offsets, sizeof(uint64_t) * (P + 1), hipHostMallocDefault));
    t->cap_pairs = P;
  }
  hipStream_t s = t->stream;
  t->count_timed = t->fill_timed = false;
  if (t->profiling && !t->events_ready) {
    for (auto& e : t->ev) TR_TRY(hipEventCreate(&e));
    t->events_ready = true;
  }
  if (t->profiling) TR_TRY(hipEventRecord(t->ev[0], s));
  hipLaunchKernelGGL(k_tracks_count, dim3(P), dim3(TR_THREADS), 0, s, dev_view(t), M, N, t->d_counts);
  TR_TRY(hipGetLastError());
  if (t->profiling) {
    TR_TRY(hipEventRecord(t->ev[1], s));
    t->count_timed = true;
  }
  TR_TRY(hipMemcpyAsync(t->h_counts, t->d_counts, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, s));
  TR_TRY(hipStreamSynchronize(s));
  t->h_offsets[0] = 0;
  for (uint32_t p = 0; p < P; ++p) t->h_offsets[p + 1] = t->h_offsets[p] + t->h_counts[p];
  {  // algorithmic bytes of the fill that follows (include/mvo_hip_tracks.h)
    uint64_t bytes = 0;
    uint32_t p = 0;
    for (uint32_t a = 0; a < M; ++a)
      for (uint32_t b = a; b + 1 < N; ++b, ++p)
        bytes += 4ull * t->h_nfeat[(t->head + t->max_frames - a) % t->max_frames] + (uint64_t)t->h_counts[p] * (96 + 4 * (b - a + 1));
    t->last_fill_bytes = bytes;
    t->last_blocks = t->h_offsets[P];
  }
  TR_TRY(hipMemcpyAsync(t->d_offsets, t->h_offsets, sizeof(uint64_t) * (P + 1), hipMemcpyHostToDevice, s));
  return MVO_HIP_OK;
}

// k_tracks_fill into m0 / m1 (device, room for h_offsets[P] blocks); queued on the stream
int fill_pairs(mvo_hip_tracks_context* t, uint32_t M, uint32_t N, double* m0, double* m1) {
  const uint32_t P = num_pairs(M, N);
  if (t->h_offsets[P] == 0) return MVO_HIP_OK;
  if (t->profiling) TR_TRY(hipEventRecord(t->ev[2], t->stream));
  hipLaunchKernelGGL(k_tracks_fill, dim3(P), dim3(TR_THREADS), 0, t->stream, dev_view(t), M, N, t->d_offsets, m0, m1);
  TR_TRY(hipGetLastError());
  if (t->profiling) {
    TR_TRY(hipEventRecord(t->ev[3], t->stream));
    t->fill_timed = true;
  }
  return MVO_HIP_OK;
}

}  // namespace

extern "C" {

int mvo_hip_tracks_create(mvo_hip_handle solver, uint32_t max_frames, uint32_t max_features,
                          mvo_hip_tracks_handle* out) {
  if (!out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!solver || max_frames < 2 || max_features == 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (max_frames > MVO_HIP_MAX_FRAMES) return MVO_HIP_ERR_UNSUPPORTED;
  mvo_hip_tracks_context* t = new (std::nothrow) mvo_hip_tracks_context();
  if (!t) return MVO_HIP_ERR_OUT_OF_MEMORY;
  t->solver = solver;
  t->device = mvo_hip_internal_device(solver);
  t->stream = (hipStream_t)mvo_hip_internal_stream(solver);
  t->max_frames = max_frames;
  t->max_features = max_features;
  t->h_nfeat.assign(max_frames, 0);
  const size_t cells = (size_t)max_frames * max_features;
  const bool ok = hipSetDevice(t->device) == hipSuccess &&
                  hipMalloc((void**)&t->d_rays, sizeof(double) * 3 * cells) == hipSuccess &&
                  hipMalloc((void**)&t->d_pre, sizeof(int32_t) * cells) == hipSuccess &&
                  hipMalloc((void**)&t->d_counter, sizeof(uint32_t) * cells) == hipSuccess &&
                  hipMalloc((void**)&t->d_pre_in, sizeof(int32_t) * max_features) == hipSuccess &&
                  hipMalloc((void**)&t->d_nfeat, sizeof(uint32_t) * max_frames) == hipSuccess &&
                  hipMemset(t->d_nfeat, 0, sizeof(uint32_t) * max_frames) == hipSuccess;
  if (!ok) {
    mvo_hip_tracks_destroy(t);
    return MVO_HIP_ERR_DEVICE;
  }
  *out = t;
  return MVO_HIP_OK;
}

void mvo_hip_tracks_destroy(mvo_hip_tracks_handle t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  void* bufs[] = {t->d_rays, t->d_pre, t->d_counter, t->d_nfeat, t->d_counts, t->d_pre_in, t->d_offsets, t->d_m0, t->d_m1};
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  if (t->h_counts) (void)hipHostFree(t->h_counts);
  if (t->h_offsets) (void)hipHostFree(t->h_offsets);
  if (t->events_ready)
    for (auto& e : t->ev) (void)hipEventDestroy(e);
  delete t;
}

int mvo_hip_tracks_reset(mvo_hip_tracks_handle t) {
  if (!t) return MVO_HIP_ERR_INVALID_ARGUMENT;
  TR_TRY(hipSetDevice(t->device));
  TR_TRY(hipStreamSynchronize(t->stream));
  t->n_pushed = 0;
  t->head = 0;
  return MVO_HIP_OK;
}

int mvo_hip_tracks_push(mvo_hip_tracks_handle t, uint32_t n_features, const double* rays, const int32_t* pre_feature) {
  if (!t || (n_features && (!rays || !pre_feature))) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (n_features > t->max_features) return MVO_HIP_ERR_UNSUPPORTED;
  TR_TRY(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  const uint32_t prev = t->head;
  const uint32_t slot = t->n_pushed == 0 ? 0 : (t->head + 1) % t->max_frames;
  const uint32_t prev_n = t->n_pushed == 0 ? 0 : t->h_nfeat[prev];
  if (n_features) {
    TR_TRY(hipMemcpyAsync(t->d_rays + (size_t)slot * t->max_features * 3, rays, sizeof(double) * 3 * n_features,
                          hipMemcpyHostToDevice, s));
    TR_TRY(hipMemcpyAsync(t->d_pre_in, pre_feature, sizeof(int32_t) * n_features, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_tracks_counter, dim3((n_features + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, s,
                       n_features, t->d_pre_in, t->d_counter + (size_t)prev * t->max_features, prev_n,
                       t->d_pre + (size_t)slot * t->max_features, t->d_counter + (size_t)slot * t->max_features);
    TR_TRY(hipGetLastError());
  }
  TR_TRY(hipMemcpyAsync(t->d_nfeat + slot, &n_features, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  TR_TRY(hipStreamSynchronize(s));  // the caller's arrays (and n_features on our stack) may go away
  t->h_nfeat[slot] = n_features;
  t->head = slot;
  if (t->n_pushed < 0xFFFFFFFFu) ++t->n_pushed;
  return MVO_HIP_OK;
}

int mvo_hip_tracks_gather(mvo_hip_tracks_handle t, uint32_t M, uint32_t N, uint64_t* pair_offsets, double* m0,
                          double* m1, uint64_t capacity) {
  if (!t || !pair_offsets) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = count_pairs(t, M, N);
  if (rc != MVO_HIP_OK) return rc;
  const uint32_t P = num_pairs(M, N);
  TR_TRY(hipStreamSynchronize(t->stream));  // d_offsets copy reads pinned h_offsets
  std::memcpy(pair_offsets, t->h_offsets, sizeof(uint64_t) * (P + 1));
  const uint64_t total = t->h_offsets[P];
  if (!m0 || !m1) return MVO_HIP_OK;
  if (capacity < total) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (total == 0) return MVO_HIP_OK;
  if (total > t->cap_obs) {
    if (t->d_m0) (void)hipFree(t->d_m0);
    if (t->d_m1) (void)hipFree(t->d_m1);
    t->d_m0 = t->d_m1 = nullptr;
    t->cap_obs = 0;
    const uint64_t want = total + total / 4 + 64;
    TR_TRY(hipMalloc((void**)&t->d_m0, sizeof(double) * 3 * want));
    TR_TRY(hipMalloc((void**)&t->d_m1, sizeof(double) * 3 * want));
    t->cap_obs = want;
  }
  rc = fill_pairs(t, M, N, t->d_m0, t->d_m1);
  if (rc != MVO_HIP_OK) return rc;
  TR_TRY(hipMemcpyAsync(m0, t->d_m0, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, t->stream));
  TR_TRY(hipMemcpyAsync(m1, t->d_m1, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, t->stream));
  TR_TRY(hipStreamSynchronize(t->stream));
  return MVO_HIP_OK;
}

int mvo_hip_tracks_refine(mvo_hip_tracks_handle t, mvo_hip_frame* frames, uint32_t M, uint32_t N,
                          const mvo_hip_options* options, mvo_hip_summary* summary) {
  if (!t || !frames || !options) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = count_pairs(t, M, N);
  if (rc != MVO_HIP_OK) return rc;
  static double placeholder[3] = {0.0, 0.0, 1.0};  // never read: the rays are written on the device
  mvo_hip_observations obs;
  obs.n_pairs = num_pairs(M, N);
  obs.reserved = 0;
  obs.pair_offsets = t->h_offsets;
  obs.m0 = placeholder;
  obs.m1 = placeholder;
  double *m0 = nullptr, *m1 = nullptr;
  rc = mvo_hip_internal_upload_device_rays(t->solver, frames, M, N, options, &obs, &m0, &m1);
  if (rc != MVO_HIP_OK) return rc;
  rc = fill_pairs(t, M, N, m0, m1);  // straight into the solver's ray buffers
  if (rc != MVO_HIP_OK) return rc;
  rc = mvo_hip_solve(t->solver);
  if (rc != MVO_HIP_OK) return rc;
  return mvo_hip_download(t->solver, frames, summary);
}

int mvo_hip_tracks_set_profiling(mvo_hip_tracks_handle t, int enabled) {
  if (!t) return MVO_HIP_ERR_INVALID_ARGUMENT;
  t->profiling = enabled != 0;
  if (!t->profiling) t->count_timed = t->fill_timed = false;
  return MVO_HIP_OK;
}

int mvo_hip_tracks_get_timing(mvo_hip_tracks_handle t, mvo_hip_tracks_timing* out) {
  if (!t || !out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof(*out));
  TR_TRY(hipSetDevice(t->device));
  float ms = 0.f;
  if (t->count_timed) {
    TR_TRY(hipEventSynchronize(t->ev[1]));
    TR_TRY(hipEventElapsedTime(&ms, t->ev[0], t->ev[1]));
    out->count_seconds = ms * 1e-3;
  }
  if (t->fill_timed) {
    TR_TRY(hipEventSynchronize(t->ev[3]));
    TR_TRY(hipEventElapsedTime(&ms, t->ev[2], t->ev[3]));
    out->fill_seconds = ms * 1e-3;
  }
  out->residual_blocks = t->last_blocks;
  out->fill_bytes = t->last_fill_bytes;
  return MVO_HIP_OK;
}

}  // extern "C"

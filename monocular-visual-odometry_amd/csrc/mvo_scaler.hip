// mvo_scaler.hip -- the closed-form scale stage on gfx950: C ABI of include/mvo_hip_scaler.h.
// Replaces the arithmetic of Scaler::stage (/root/reference/mvo/src/pipeline/Scaler.cpp:21-68).
// No CPU fallback: every entry point needs a HIP device.
//
//   k_scaler_small   <= 2048 tracks (the reference's real operating point is <= 60): ONE launch,
//                    one workgroup -- per-track scales, bitonic sort of the order-preserving keys
//                    in LDS, the two middle values
//   k_scaler_tracks  per-track scales, 72 B read + 8 B written per track (HBM-bound), and the
//                    histogram of the keys' top digit while they are in registers
//   radix select of the middle rank, 11-bit digits, every kernel picks the previous digit's bin in
//   its prologue (no separate pick launches, no host round trips):
//   k_scaler_hist<true>   digit 1 over all keys (8 B per track)
//   k_scaler_compact      keys that share the 22-bit prefix of the wanted rank -> candidate list
//                         (typically n / 1000 keys), digit 2's histogram on the way, and the
//                         smallest key above the bucket
//   k_scaler_hist<false>  digits 3..5 over the candidates only
//   k_scaler_final        last pick = the lower middle key; for an even count the number of
//                         candidates <= it and the smallest one above it
//   Histograms are per-workgroup in LDS (wave-aggregated adds: clustered keys would otherwise
//   serialise on one bin) and merged with integer atomics: order-independent, deterministic.
#include "../../include/mvo_hip_scaler.h"
#include "../../include/mvo_hip.h"
#include "mvo_internal.h"

#include <hip/hip_runtime.h>

#include "mvo_alloc.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#define SC_SMALL 2048
#define SC_BINS 2048

struct ScalerIn {  // == mvo_hip_scaler_input
  double R0[9], R1[9], R2[9], u1[3], n1, lowest, highest;
};
static_assert(sizeof(ScalerIn) == sizeof(mvo_hip_scaler_input), "input record layout");

#define SC_PASSES 6
#define SC_STAGE 2048

struct ScalerSel {
  unsigned long long prefix, mask, rank;  // keys with (key & mask) == prefix are still candidates
};

struct ScalerCtl {
  // set by the host before every run
  ScalerSel sel0;
  unsigned long long outside_min, min_gt;  // start at ~0
  // zeroed before every run
  ScalerSel sel[SC_PASSES];                // sel[p]: after digits 0..p picked (written by workgroup 0)
  unsigned long long n_cand, count_le, key1, key2;
  unsigned long long guess0, use_guess;    // k_scaler_tracks' guess of the picked digit 0; it was right
  uint32_t hist[SC_PASSES][SC_BINS];
  uint32_t hist1g[SC_BINS];                // digit 1 over the keys whose digit 0 is guess0
};
#define SC_CTL_INIT_BYTES 40
#define SC_CTL_RESULT_BYTES (SC_CTL_INIT_BYTES + sizeof(ScalerSel) * SC_PASSES + 48)

__host__ __device__ __forceinline__ int sc_shift(int p) { return p < 5 ? 53 - 11 * p : 0; }
__host__ __device__ __forceinline__ int sc_bits(int p) { return p < 5 ? 11 : 9; }

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
                                                      ScalerCtl* __restrict__ ctl) {
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
    ctl->key1 = key[(n - 1) / 2];
    ctl->key2 = (n % 2 == 0) ? key[(n - 1) / 2 + 1] : key[(n - 1) / 2];
  }
}

// LDS histogram add for one wavefront's keys: two rounds of "all lanes with the leader's digit add
// once", then plain atomics for the rest.  Must be reached by the whole wavefront.
__device__ __forceinline__ void sc_hist_add(uint32_t* lh, uint32_t digit, bool valid) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (!todo) break;
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader)  /* leader is wave-uniform: no trip through the LDS crossbar */;
    const unsigned long long same = __ballot(valid && digit == d) & todo;
    if (lane == leader) atomicAdd(&lh[d], (uint32_t)__popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1ull) atomicAdd(&lh[digit], 1u);
}

__device__ __forceinline__ void sc_hist_merge(const uint32_t* lh, uint32_t* __restrict__ hist) {
  __syncthreads();
  for (int b = threadIdx.x; b < SC_BINS; b += 256)
    if (lh[b]) atomicAdd(&hist[b], lh[b]);
}

// The same with SC_COPIES histograms side by side (lane & (SC_COPIES - 1) picks one): digits of real
// data cluster in a few dozen bins, and lanes that hit one LDS address in one instruction are served
// one after the other.  One aggregation round takes out the most frequent digit first.
#define SC_COPIES 4
#define SC_COPIES_TRACKS 2  // k_scaler_tracks: 24 KB of LDS in all, six workgroups per CU
template <int COPIES = SC_COPIES>
__device__ __forceinline__ void sc_hist_add_rep(uint32_t* lh, uint32_t digit, bool valid) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  if (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader)  /* leader is wave-uniform: no trip through the LDS crossbar */;
    const unsigned long long same = __ballot(valid && digit == d);
    if (lane == leader) atomicAdd(&lh[d], (uint32_t)__popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1ull) atomicAdd(&lh[(lane & (COPIES - 1)) * SC_BINS + digit], 1u);
}

template <int COPIES = SC_COPIES>
__device__ __forceinline__ void sc_hist_merge_rep(const uint32_t* lh, uint32_t* __restrict__ hist) {
  __syncthreads();
  for (int b = threadIdx.x; b < SC_BINS; b += 256) {
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < COPIES; ++q) c += lh[q * SC_BINS + b];
    if (c) atomicAdd(&hist[b], c);
  }
}

// Every thread of a 256-thread workgroup: the state after picking digit p from its histogram.
__device__ ScalerSel sc_pick(ScalerSel prev, const uint32_t* __restrict__ hist, int p) {
  __shared__ unsigned long long wave_total[4];
  __shared__ ScalerSel picked;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = 1 << sc_bits(p), shift = sc_shift(p);
  uint32_t c[8];
  unsigned long long local = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int b = tid * 8 + q;
    c[q] = b < nb ? hist[b] : 0u;
    local += c[q];
  }
  unsigned long long incl = local;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  for (int w = 0; w < wave; ++w) incl += wave_total[w];
  const unsigned long long excl = incl - local, rank = prev.rank;
  if (rank >= excl && rank < incl) {  // exactly one thread
    unsigned long long below = excl;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (rank < below + c[q]) {
        picked.prefix = prev.prefix | ((unsigned long long)(tid * 8 + q) << shift);
        picked.mask = prev.mask | (((1ull << sc_bits(p)) - 1) << shift);
        picked.rank = rank - below;
        break;
      }
      below += c[q];
    }
  }
  __syncthreads();
  return picked;
}

// Also histograms digit 0 and, speculatively, digit 1 of the keys whose digit 0 equals a guess: the
// digit 0 of the median of the first 64 tracks (every workgroup derives the same guess).  When the
// select then picks that very bin for digit 0 -- it does unless the first 64 tracks are unlike the
// rest -- digit 1's histogram is already there and the separate pass over the keys is skipped
// (k_scaler_hist<true> checks and returns); otherwise nothing is lost but that shortcut.
__global__ __launch_bounds__(256) void k_scaler_tracks(ScalerIn in, uint64_t n, const double* __restrict__ v0,
                                                      const double* __restrict__ v1,
                                                      const double* __restrict__ v2, double* __restrict__ scales,
                                                      ScalerCtl* __restrict__ ctl) {
  __shared__ uint32_t lh[SC_BINS];
  __shared__ uint32_t lh1[SC_COPIES_TRACKS * SC_BINS];
  __shared__ uint32_t guess_sh;
  for (int b = threadIdx.x; b < SC_BINS; b += 256) lh[b] = 0;
  for (int b = threadIdx.x; b < SC_COPIES_TRACKS * SC_BINS; b += 256) lh1[b] = 0;
  if (threadIdx.x < 64) {  // wavefront 0: rank of every lane's key among the first min(64, n)
    const uint32_t lane = threadIdx.x, m = n < 64 ? (uint32_t)n : 64u;
    unsigned long long k = ~0ull;
    if (lane < m) k = sc_key(sc_track(in, v0 + 3 * lane, v1 + 3 * lane, v2 + 3 * lane));
    uint32_t below = 0;
    for (uint32_t j = 0; j < m; ++j) {
      const unsigned long long o = __shfl(k, (int)j, 64);
      below += (o < k || (o == k && j < lane)) ? 1u : 0u;
    }
    if (lane < m && below == (m - 1) / 2) guess_sh = (uint32_t)(k >> sc_shift(0));
  }
  __syncthreads();
  const uint32_t guess = guess_sh;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->guess0 = guess;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t base = (uint64_t)blockIdx.x * 256; base < n; base += stride) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n;
    unsigned long long k = 0;
    if (valid) {
      const double s = sc_track(in, v0 + 3 * i, v1 + 3 * i, v2 + 3 * i);
      scales[i] = s;
      k = sc_key(s);
    }
    const uint32_t d0 = (uint32_t)(k >> sc_shift(0));
    sc_hist_add(lh, d0, valid);
    sc_hist_add_rep<SC_COPIES_TRACKS>(lh1, (uint32_t)((k >> sc_shift(1)) & ((1u << sc_bits(1)) - 1)), valid && d0 == guess);
  }
  sc_hist_merge(lh, ctl->hist[0]);
  sc_hist_merge_rep<SC_COPIES_TRACKS>(lh1, ctl->hist1g);
}

// histogram of digit p over the keys that match the state after digit p - 1 (picked here)
template <bool FROM_SCALES>
__global__ __launch_bounds__(256) void k_scaler_hist(uint64_t n, const double* __restrict__ scales,
                                                    const unsigned long long* __restrict__ cand,
                                                    ScalerCtl* __restrict__ ctl, int p) {
  __shared__ uint32_t lh[SC_COPIES * SC_BINS];
  const uint64_t count = FROM_SCALES ? n : ctl->n_cand;
  if (blockIdx.x != 0 && (uint64_t)blockIdx.x * 1024 >= count) return;
  const ScalerSel S = sc_pick(p == 1 ? ctl->sel0 : ctl->sel[p - 2], ctl->hist[p - 1], p - 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->sel[p - 1] = S;
  if (FROM_SCALES) {  // p == 1: k_scaler_tracks guessed digit 0's bin right -> its histogram serves
    const bool hit = (S.prefix >> sc_shift(0)) == ctl->guess0;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->use_guess = hit ? 1ull : 0ull;
    if (hit) return;
  }
  for (int b = threadIdx.x; b < SC_COPIES * SC_BINS; b += 256) lh[b] = 0;
  __syncthreads();
  const int shift = sc_shift(p);
  const unsigned long long dm = (1ull << sc_bits(p)) - 1;
  const uint64_t stride = (uint64_t)gridDim.x * 1024;
  for (uint64_t base = (uint64_t)blockIdx.x * 1024; base < count; base += stride) {
    unsigned long long k[4];
    bool valid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // four loads in flight per thread
      const uint64_t i = base + j * 256 + threadIdx.x;
      valid[j] = i < count;
      // unconditional, at a clamped index (count > 0 inside this loop): a load inside a divergent `if`
      // is waited for on the spot, which serialises the four
      const uint64_t ic = valid[j] ? i : count - 1;
      k[j] = FROM_SCALES ? sc_key(scales[ic]) : cand[ic];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      sc_hist_add_rep(lh, (uint32_t)((k[j] >> shift) & dm), valid[j] && (k[j] & S.mask) == S.prefix);
  }
  sc_hist_merge_rep(lh, ctl->hist[p]);
}

// picks digit 1; keys in the wanted rank's 22-bit bucket -> cand (order irrelevant: they are only
// counted and compared), digit 2's histogram, smallest key above the bucket
__global__ __launch_bounds__(256) void k_scaler_compact(uint64_t n, const double* __restrict__ scales,
                                                       unsigned long long* __restrict__ cand,
                                                       ScalerCtl* __restrict__ ctl) {
  __shared__ uint32_t lh[SC_BINS];
  __shared__ unsigned long long stage[SC_STAGE];
  __shared__ uint32_t n_stage;
  __shared__ unsigned long long g_base, wave_min[4];
  const ScalerSel S = sc_pick(ctl->sel[0], ctl->use_guess ? ctl->hist1g : ctl->hist[1], 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->sel[1] = S;
  for (int b = threadIdx.x; b < SC_BINS; b += 256) lh[b] = 0;
  if (threadIdx.x == 0) n_stage = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, shift = sc_shift(2);
  const unsigned long long dm = (1ull << sc_bits(2)) - 1;
  unsigned long long above = ~0ull;
  const uint64_t stride = (uint64_t)gridDim.x * 1024;
  auto flush = [&]() {  // whole workgroup
    if (threadIdx.x == 0) g_base = atomicAdd(&ctl->n_cand, (unsigned long long)n_stage);
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < n_stage; q += 256) cand[g_base + q] = stage[q];
    __syncthreads();
    if (threadIdx.x == 0) n_stage = 0;
    __syncthreads();
  };
  for (uint64_t base = (uint64_t)blockIdx.x * 1024; base < n; base += stride) {
    unsigned long long k[4];
    bool valid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // four loads in flight per thread
      const uint64_t i = base + j * 256 + threadIdx.x;
      valid[j] = i < n;
      k[j] = sc_key(scales[valid[j] ? i : n - 1]);  // unconditional load at a clamped index (n > 0 here)
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long top = k[j] & S.mask;
      const bool match = valid[j] && top == S.prefix;
      if (valid[j] && top > S.prefix && k[j] < above) above = k[j];
      const unsigned long long bal = __ballot(match);
      if (bal) {  // wavefront-uniform
        const int leader = __ffsll((long long)bal) - 1;
        uint32_t wpos = 0;
        if (lane == leader) wpos = atomicAdd(&n_stage, (uint32_t)__popcll(bal));
        wpos = (uint32_t)__builtin_amdgcn_readlane((int)wpos, leader);
        if (match) stage[wpos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = k[j];
        sc_hist_add(lh, (uint32_t)((k[j] >> shift) & dm), match);
      }
    }
    __syncthreads();
    if (n_stage > SC_STAGE - 1024) flush();  // workgroup-uniform: n_stage is read after the barrier
  }
  if (n_stage) flush();
  sc_hist_merge(lh, ctl->hist[2]);
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(above, o, 64);
    above = t < above ? t : above;
  }
  if (lane == 0) wave_min[threadIdx.x >> 6] = above;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long m = wave_min[0];
    for (int q = 1; q < 4; ++q) m = wave_min[q] < m ? wave_min[q] : m;
    if (m != ~0ull) atomicMin(&ctl->outside_min, m);
  }
}

// last pick = the lower middle key; candidates <= it and the smallest candidate above it
__global__ __launch_bounds__(256) void k_scaler_final(const unsigned long long* __restrict__ cand,
                                                     ScalerCtl* __restrict__ ctl) {
  __shared__ unsigned long long wc[4], wm[4];
  const uint64_t count = ctl->n_cand;
  if (blockIdx.x != 0 && (uint64_t)blockIdx.x * 256 >= count) return;
  const ScalerSel S = sc_pick(ctl->sel[SC_PASSES - 2], ctl->hist[SC_PASSES - 1], SC_PASSES - 1);
  const unsigned long long k1 = S.prefix;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->sel[SC_PASSES - 1] = S;
    ctl->key1 = k1;
  }
  unsigned long long cnt = 0, mn = ~0ull;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const unsigned long long k = cand[i];
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
    if (c) atomicAdd(&ctl->count_le, c);
    if (m != ~0ull) atomicMin(&ctl->min_gt, m);
  }
}

struct mvo_hip_scaler_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipError_t last_hip_error = hipSuccess;
  double *d_v[3] = {nullptr, nullptr, nullptr}, *d_scales = nullptr;
  uint64_t cap = 0, n = 0;
  unsigned long long* d_cand = nullptr;  // keys of the wanted rank's 22-bit bucket (room for all)
  ScalerCtl* d_ctl = nullptr;
  ScalerCtl* h_ctl = nullptr;  // pinned; only the init and result parts are ever copied
  hipEvent_t ev[4] = {};       // whole run; k_scaler_tracks
  double device_seconds = 0.0, tracks_seconds = 0.0;
  uint64_t bytes = 0, tracks_bytes = 0;
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
    if (p) (void)mvo_device_free(p);
    p = nullptr;
  }
  if (h->d_scales) (void)mvo_device_free(h->d_scales);
  if (h->d_cand) (void)mvo_device_free(h->d_cand);
  h->d_scales = nullptr;
  h->d_cand = nullptr;
  h->cap = 0;
  const uint64_t want = n + n / 8 + 64;
  for (auto& p : h->d_v) SC_TRY(mvo_device_alloc((void**)&p, sizeof(double) * 3 * want));
  SC_TRY(mvo_device_alloc((void**)&h->d_scales, sizeof(double) * want));
  SC_TRY(mvo_device_alloc((void**)&h->d_cand, sizeof(unsigned long long) * want));
  h->cap = want;
  return MVO_HIP_OK;
}

int sc_upload(mvo_hip_scaler_context* h, uint64_t n, const double* v0, const double* v1, const double* v2) {
  if (n && (!v0 || !v1 || !v2)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (n >= (1ull << 32)) return MVO_HIP_ERR_UNSUPPORTED;  // 32-bit histogram bins
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
  h->tracks_seconds = 0.0;
  h->tracks_bytes = 0;
  if (n <= SC_SMALL) {
    hipLaunchKernelGGL(k_scaler_small, dim3(1), dim3(1024), 0, s, ki, (uint32_t)n, h->d_v[0], h->d_v[1], h->d_v[2],
                       h->d_scales, h->d_ctl);
    SC_TRY(hipGetLastError());
    h->bytes = 80ull * n;
  } else {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, std::min<uint64_t>((uint64_t)h->n_cu * 16, 8192));
    uint32_t per_cu = 4;
    if (const char* e = std::getenv("MVO_SC_BLOCKS_PER_CU")) per_cu = (uint32_t)std::max(1, std::atoi(e));  // tuning knob
    const uint32_t cand_blocks = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)h->n_cu * per_cu);
    h->h_ctl->sel0.prefix = 0;
    h->h_ctl->sel0.mask = 0;
    h->h_ctl->sel0.rank = (n - 1) / 2;
    h->h_ctl->outside_min = ~0ull;
    h->h_ctl->min_gt = ~0ull;
    SC_TRY(hipMemcpyAsync(h->d_ctl, h->h_ctl, SC_CTL_INIT_BYTES, hipMemcpyHostToDevice, s));
    SC_TRY(hipMemsetAsync(reinterpret_cast<char*>(h->d_ctl) + SC_CTL_INIT_BYTES, 0, sizeof(ScalerCtl) - SC_CTL_INIT_BYTES, s));
    SC_TRY(hipEventRecord(h->ev[2], s));
    uint32_t tracks_blocks = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)h->n_cu * 6);
    if (const char* e = std::getenv("MVO_SC_TRACKS_BLOCKS_PER_CU"))
      tracks_blocks = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)h->n_cu * std::max(1, std::atoi(e)));  // tuning knob
    hipLaunchKernelGGL(k_scaler_tracks, dim3(tracks_blocks), dim3(256), 0, s, ki, n, h->d_v[0], h->d_v[1], h->d_v[2],
                       h->d_scales, h->d_ctl);
    SC_TRY(hipEventRecord(h->ev[3], s));
    // few, long-lived workgroups for the passes over the keys: the per-workgroup prologue (pick) and
    // epilogue (histogram merge, appends) are paid n_cu * 4 times, not once per 1024 keys
    hipLaunchKernelGGL(k_scaler_hist<true>, dim3(cand_blocks), dim3(256), 0, s, n, h->d_scales, h->d_cand, h->d_ctl, 1);
    hipLaunchKernelGGL(k_scaler_compact, dim3(cand_blocks), dim3(256), 0, s, n, h->d_scales, h->d_cand, h->d_ctl);
    for (int p = 3; p < SC_PASSES; ++p)
      hipLaunchKernelGGL(k_scaler_hist<false>, dim3(cand_blocks), dim3(256), 0, s, n, h->d_scales, h->d_cand, h->d_ctl, p);
    hipLaunchKernelGGL(k_scaler_final, dim3(cand_blocks), dim3(256), 0, s, h->d_cand, h->d_ctl);
    SC_TRY(hipGetLastError());
    h->tracks_bytes = 80ull * n;
  }
  SC_TRY(hipEventRecord(h->ev[1], s));
  SC_TRY(hipMemcpyAsync(h->h_ctl, h->d_ctl, SC_CTL_RESULT_BYTES, hipMemcpyDeviceToHost, s));
  if (per_track) SC_TRY(hipMemcpyAsync(per_track, h->d_scales, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  SC_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  SC_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->device_seconds = ms * 1e-3;
  if (n > SC_SMALL) {
    SC_TRY(hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
    h->tracks_seconds = ms * 1e-3;
    // rays in, scales out, one pass over the keys (two when the digit-0 guess missed)
    h->bytes = (72ull + 8 + (h->h_ctl->use_guess ? 8 : 16)) * n;
  }
  unsigned long long k1 = h->h_ctl->key1, k2 = h->h_ctl->key2;
  if (n > SC_SMALL) {
    // the upper middle of an even count: candidates are ALL keys of key1's 22-bit bucket, the keys
    // below the bucket number (n - 1) / 2 - sel[1].rank, so "(n - 1) / 2 + 2 keys are <= key1"
    // reads count_le >= sel[1].rank + 2; otherwise it is the smallest key above key1, inside the
    // bucket or above it
    const unsigned long long up = std::min(h->h_ctl->min_gt, h->h_ctl->outside_min);
    k2 = (n % 2 == 0) ? (h->h_ctl->count_le >= h->h_ctl->sel[1].rank + 2 ? k1 : up) : k1;
  }
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
            mvo_device_alloc((void**)&h->d_ctl, sizeof(ScalerCtl)) == hipSuccess &&
            hipHostMalloc((void**)&h->h_ctl, sizeof(ScalerCtl), hipHostMallocDefault) == hipSuccess &&
            hipEventCreate(&h->ev[0]) == hipSuccess && hipEventCreate(&h->ev[1]) == hipSuccess &&
            hipEventCreate(&h->ev[2]) == hipSuccess && hipEventCreate(&h->ev[3]) == hipSuccess;
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
    if (p) (void)mvo_device_free(p);
  if (h->d_scales) (void)mvo_device_free(h->d_scales);
  if (h->d_cand) (void)mvo_device_free(h->d_cand);
  if (h->d_ctl) (void)mvo_device_free(h->d_ctl);
  if (h->h_ctl) (void)hipHostFree(h->h_ctl);
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

// library-internal (csrc/mvo_internal.h): the device-side three-frame gather of mvo_tracks.hip writes the
// rays straight into this handle's buffers
int mvo_hip_scaler_internal_reserve(mvo_hip_scaler_handle h, uint64_t n_tracks, double** v0, double** v1, double** v2,
                                    int* device) {
  if (!h || !v0 || !v1 || !v2) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (n_tracks >= (1ull << 32)) return MVO_HIP_ERR_UNSUPPORTED;
  SC_TRY(hipSetDevice(h->device));
  SC_TRY(hipStreamSynchronize(h->stream));  // a run may still be reading the buffers about to be replaced
  const int rc = sc_reserve(h, n_tracks ? n_tracks : 1);
  if (rc != MVO_HIP_OK) return rc;
  h->n = n_tracks;
  *v0 = h->d_v[0];
  *v1 = h->d_v[1];
  *v2 = h->d_v[2];
  if (device) *device = h->device;
  return MVO_HIP_OK;
}

int mvo_hip_scaler_tracks_timing(mvo_hip_scaler_handle h, double* device_seconds, uint64_t* bytes) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (device_seconds) *device_seconds = h->tracks_seconds;
  if (bytes) *bytes = h->tracks_bytes;
  return MVO_HIP_OK;
}

}  // extern "C"

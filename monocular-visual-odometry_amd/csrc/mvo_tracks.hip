// mvo_tracks.hip -- device-resident feature-track tables and the on-device observation gather:
// C ABI of include/mvo_hip_tracks.h.  Replaces the host-side chain walk of
// IterativeRefinement::refine (/root/reference/mvo/src/algorithms/IterativeRefinement.cpp:116-165,
// Frame::getCorrespondingFeatures sliding_window/Frame.hpp:258-302, normFeatures
// operations/FeatureOperations.cpp:117-122).  No CPU fallback.
//
// The reference walks the predecessor chain of a feature again for every frame pair it takes part
// in (b - a + 1 steps for pair (a, b): quadratic in the track length).  Here a feature of frame a
// walks its chain ONCE and drops one residual block per step -- step s is pair (a, a + s - 1) --
// at the position the reference's loop would have pushed it: pair offset + number of features
// with a smaller index in frame a that also reach back s frames (Frame.hpp:277-291 keeps feature
// order).  Those ranks come from a count + scan over 64-feature groups:
//
//   k_tracks_counter  on push: _preFeatureCounter of the new frame (Frame.cpp:132-148) and the unit
//                     ray of every feature (normFeatures is a pure function of the stored ray, so
//                     normalising once at push returns the very same doubles)
//   k_tracks_count    wavefront per (64-feature group g, frame a): cnt[a][g][s] = features of the
//                     group that reach back s frames
//   k_tracks_scan     per frame a and step s: exclusive scan of cnt over the groups (-> the group's
//                     base rank, in place) and the pair's block count
//   (host: prefix sum of the P pair counts -> CSR offsets, needed by the solver's tiling anyway)
//   k_tracks_fill     wavefront per (group, frame a, 16-step segment): the chain walk; ranks inside the
//                     group by ballot; m0 / m1 written straight into the solver's ray buffers
//   k_tracks_skip     on push: skip links of the new frame (the feature 16 / 32 / 64 / 128 frames back
//                     along the track), so that a walker can START at step 16 k after at most four
//                     lookups: the walk is a chain of dependent loads (~0.7 us each from the 256-MB
//                     cache), and 199 of them in a row were all of k_tracks_fold's 143 us at N = 200
// All kernels run on the solver's stream; the solve is queued right behind the gather.
#include "../../include/mvo_hip_tracks.h"

#include <hip/hip_runtime.h>

#include "mvo_alloc.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mvo_internal.h"
#include "mvo_wave.h"

#define MVO_FRAME_DOUBLES 13  // R(9) vec(3) scale(1): mvo_hip_frame

#define TR_THREADS 256
#define TR_SCAN_GROUPS 16  // k_tracks_scan: 16 wavefronts share the feature groups of one (a, 64 steps)
#define TR_SEG 16          // steps one walker wavefront takes (k_tracks_fill / k_tracks_fold)
#define TR_SKIP_LEVELS 4   // skip links of TR_SEG << level frames: 16, 32, 64, 128 (N <= MVO_HIP_MAX_FRAMES = 256)
static_assert((TR_SEG << TR_SKIP_LEVELS) >= MVO_HIP_MAX_FRAMES, "a segment start must be reachable by the skip levels");

struct TracksDev {
  const double* rays;     // [max_frames][max_features][3], unit length
  const int32_t* pre;     // [max_frames][max_features]
  const int32_t* skip;    // [max_frames][TR_SKIP_LEVELS][max_features]: the track's feature TR_SEG << level frames back, -1 = none
  const uint32_t* counter;
  const uint32_t* n_feat; // [max_frames]
  uint32_t max_frames, max_features, head;  // slot of the newest frame
};

__device__ __forceinline__ uint32_t tr_slot(const TracksDev& t, uint32_t age) {  // age 0 = newest
  return (t.head + t.max_frames - age) % t.max_frames;
}

// slot of the next older keyframe (the ring walked one step back: no division per step)
__device__ __forceinline__ uint32_t tr_older(const TracksDev& t, uint32_t slot) {
  return slot ? slot - 1 : t.max_frames - 1;
}

// index of pair (a, a) in the reference's loop order: for a < M: for b in [a, N-1)
__host__ __device__ __forceinline__ uint32_t tr_pair_start(uint32_t N, uint32_t a) {
  return a * (N - 1) - (a * (a - 1)) / 2;
}

// FeatureOperations.cpp:119: scale = 1.0 / cv::norm(v).  Contraction is switched off here: a fused
// multiply-add in the sum of squares rounds differently from the host's separate multiply and add
// (the _rn intrinsics are plain operators to the compiler and would be fused all the same).
__device__ __forceinline__ double tr_inv_norm(double x, double y, double z) {
#pragma clang fp contract(off)
  const double xx = x * x, yy = y * y, zz = z * z;
  const double ss = (xx + yy) + zz;
  return 1.0 / __builtin_sqrt(ss);
}

__global__ __launch_bounds__(TR_THREADS) void k_tracks_counter(uint32_t n, const double* __restrict__ rays_in,
                                                              const int32_t* __restrict__ pre,
                                                              const uint32_t* __restrict__ prev_counter,
                                                              uint32_t prev_n, double* __restrict__ rays_out,
                                                              int32_t* __restrict__ pre_out,
                                                              uint32_t* __restrict__ counter_out) {
  const uint32_t i = blockIdx.x * TR_THREADS + threadIdx.x;
  if (i >= n) return;
  const int32_t p = pre ? pre[i] : -1;  // no predecessor array: features appended to a frame (all new tracks)
  if (p >= 0 && (uint32_t)p < prev_n) {  // Frame.cpp:138-139
    counter_out[i] = prev_counter[p] + 1;
    pre_out[i] = p;
  } else {  // Frame.cpp:141-144
    counter_out[i] = 0;
    pre_out[i] = -1;
  }
  const double x = rays_in[3 * i], y = rays_in[3 * i + 1], z = rays_in[3 * i + 2];
  const double sc = tr_inv_norm(x, y, z);  // FeatureOperations.cpp:119-120
  rays_out[3 * i] = x * sc;
  rays_out[3 * i + 1] = y * sc;
  rays_out[3 * i + 2] = z * sc;
}

// Skip links of the frame just pushed (t.head = its slot; n_avail = frames in the ring, this one
// included).  Level 0 follows TR_SEG predecessor links; level l is two jumps of level l - 1, the second
// one read from the older frame's own links.  A link over more frames than the ring holds, or than the
// track is long, is -1.  Every index read is range-checked before it is used as an index again.
__global__ __launch_bounds__(TR_THREADS) void k_tracks_skip(TracksDev t, uint32_t first, uint32_t n, uint32_t n_avail,
                                                           int32_t* __restrict__ skip) {
  const uint32_t i = first + blockIdx.x * TR_THREADS + threadIdx.x;
  if (i >= n) return;
  const uint32_t mf = t.max_features, slot0 = t.head;
  const uint32_t c = t.counter[(size_t)slot0 * mf + i];
  bool ok = c >= TR_SEG && TR_SEG < n_avail;
  uint32_t idx = i;
  if (ok) {
    for (uint32_t s = 0; s < TR_SEG; ++s) {
      const int32_t v = t.pre[(size_t)tr_slot(t, s) * mf + idx];
      if ((uint32_t)v >= mf) {
        ok = false;
        break;
      }
      idx = (uint32_t)v;
    }
  }
  skip[((size_t)slot0 * TR_SKIP_LEVELS) * mf + i] = ok ? (int32_t)idx : -1;
  for (uint32_t l = 1; l < TR_SKIP_LEVELS; ++l) {
    const uint32_t jump = (uint32_t)TR_SEG << l;
    ok = ok && c >= jump && jump < n_avail;
    if (ok) {
      const int32_t v = skip[((size_t)tr_slot(t, jump / 2) * TR_SKIP_LEVELS + (l - 1)) * mf + idx];
      if ((uint32_t)v >= mf) ok = false;
      else idx = (uint32_t)v;
    }
    skip[((size_t)slot0 * TR_SKIP_LEVELS + l) * mf + i] = ok ? (int32_t)idx : -1;
  }
}

// index, in frame a + TR_SEG k, of the feature the track of feature i of frame a passes through
// (k wave-uniform, < 1 << TR_SKIP_LEVELS); -1 when a link is missing
__device__ __forceinline__ int32_t tr_seek(const TracksDev& t, uint32_t a, uint32_t k, uint32_t i) {
  int32_t idx = (int32_t)i;
  uint32_t age = a;
#pragma unroll
  for (int l = TR_SKIP_LEVELS - 1; l >= 0; --l) {
    if ((k >> l) & 1u) {
      if (idx >= 0) {
        const int32_t v = t.skip[((size_t)tr_slot(t, age) * TR_SKIP_LEVELS + (uint32_t)l) * t.max_features + (uint32_t)idx];
        idx = (uint32_t)v < t.max_features ? v : -1;
      }
      age += (uint32_t)TR_SEG << l;
    }
  }
  return idx;
}

// cnt[(a * n_groups + g) * SP + (s - 1)], SP = N - 1 rounded up to 64 (steps beyond N - 1 - a count 0)
__global__ __launch_bounds__(64) void k_tracks_count(TracksDev t, uint32_t N, uint32_t n_groups, uint32_t SP,
                                                    uint32_t* __restrict__ cnt) {
  const uint32_t g = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
  const uint32_t slot = tr_slot(t, a), n = t.n_feat[slot];
  if (g * 64 >= n) return;  // k_tracks_scan never reads the groups past the frame's last feature
  const uint32_t i = g * 64 + lane;
  const uint32_t c = i < n ? min(t.counter[(size_t)slot * t.max_features + i], N - 1 - a) : 0u;
  uint32_t* __restrict__ row = cnt + (size_t)(a * n_groups + g) * SP;
  for (uint32_t s0 = 0; s0 < SP; s0 += 64) {
    uint32_t mine = 0;
    for (uint32_t j = 0; j < 64; ++j) {
      const uint32_t k = (uint32_t)__popcll(__ballot(c >= s0 + j + 1));
      if (lane == j) mine = k;
    }
    row[s0 + lane] = mine;
  }
}

// grid (SP / 64, M), TR_SCAN_GROUPS wavefronts: lane <-> step, the wavefronts split the feature groups
__global__ __launch_bounds__(64 * TR_SCAN_GROUPS) void k_tracks_scan(TracksDev t, uint32_t N, uint32_t n_groups,
                                                                    uint32_t SP, uint32_t* __restrict__ cnt,
                                                                    uint32_t* __restrict__ counts) {
  __shared__ uint32_t part[TR_SCAN_GROUPS][64];
  const uint32_t a = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t s1 = blockIdx.x * 64 + lane;  // s - 1
  const uint32_t ng = (t.n_feat[tr_slot(t, a)] + 63) / 64;
  const uint32_t per = (ng + TR_SCAN_GROUPS - 1) / TR_SCAN_GROUPS;
  const uint32_t lo = min(w * per, ng), hi = min(lo + per, ng);
  uint32_t* __restrict__ col = cnt + (size_t)a * n_groups * SP + s1;
  uint32_t sum = 0;
  for (uint32_t g = lo; g < hi; ++g) sum += col[(size_t)g * SP];
  part[w][lane] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
  for (uint32_t k = 0; k < TR_SCAN_GROUPS; ++k) {
    const uint32_t v = part[k][lane];
    run += k < w ? v : 0u;
    total += v;
  }
  for (uint32_t g = lo; g < hi; ++g) {
    const uint32_t v = col[(size_t)g * SP];
    col[(size_t)g * SP] = run;
    run += v;
  }
  if (w == 0 && s1 < N - 1 - a) counts[tr_pair_start(N, a) + s1] = total;
}

// grid (groups, M, segments): the wavefront takes the steps seg * blockIdx.z + 1 ... seg * (blockIdx.z + 1)
// of its 64 features' walks (seg = TR_SEG with skip links, N - 1 and one segment without).  What it writes
// for a step depends on the step alone, so the split changes no byte of the output.
__global__ __launch_bounds__(64) void k_tracks_fill(TracksDev t, uint32_t N, uint32_t n_groups, uint32_t SP,
                                                   uint32_t seg, const uint32_t* __restrict__ base,
                                                   const uint64_t* __restrict__ offsets, double* __restrict__ m0,
                                                   double* __restrict__ m1) {
  const uint32_t g = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
  const uint32_t slot_a = tr_slot(t, a), n = t.n_feat[slot_a];
  if (g * 64 >= n) return;
  const uint32_t i = g * 64 + lane;
  const bool have = i < n;
  const uint32_t s_lo = blockIdx.z * seg;
  uint32_t c = have ? min(t.counter[(size_t)slot_a * t.max_features + i], N - 1 - a) : 0u;
  if (c <= s_lo) c = 0;  // nothing to do in this segment
  if (!__ballot(c != 0)) return;
  int32_t cur = (int32_t)i;
  if (s_lo && c) {
    cur = tr_seek(t, a, blockIdx.z, i);
    if (cur < 0) c = 0;
  }
  double r0x = 0.0, r0y = 0.0, r0z = 0.0;
  if (c) {
    const double* __restrict__ r0 = t.rays + ((size_t)slot_a * t.max_features + i) * 3;
    r0x = r0[0];
    r0y = r0[1];
    r0z = r0[2];
  }
  const uint32_t* __restrict__ brow = base + (size_t)(a * n_groups + g) * SP;
  const uint64_t* __restrict__ orow = offsets + tr_pair_start(N, a);
  const unsigned long long below = (1ull << lane) - 1ull;
  // Frame.hpp:283-289: follow _preFeature; idx is the feature's index in frame a + s - 1
  uint32_t idx = (uint32_t)cur;
  uint32_t slot_s = tr_slot(t, a + s_lo);  // wave-uniform: the keyframe the walk stands in
  int32_t nxt = c ? t.pre[(size_t)slot_s * t.max_features + idx] : -1;
  const uint32_t s_hi = s_lo + seg;
  for (uint32_t s = s_lo + 1; s <= s_hi; ++s) {
    const bool alive = c >= s;
    const unsigned long long bal = __ballot(alive);
    if (!bal) break;
    slot_s = tr_older(t, slot_s);
    if (alive) {
      idx = min((uint32_t)nxt, t.max_features - 1);  // a valid link is in range; the clamp only keeps a broken table from faulting
      const size_t cell = (size_t)slot_s * t.max_features + idx;
      const double* __restrict__ r1 = t.rays + cell * 3;
      const double x = r1[0], y = r1[1], z = r1[2];
      if (c > s) nxt = t.pre[cell];  // in flight together with the ray
      const size_t o = 3 * (size_t)(orow[s - 1] + brow[s - 1] + (uint32_t)__popcll(bal & below));
      m0[o] = r0x;
      m0[o + 1] = r0y;
      m0[o + 2] = r0z;
      m1[o] = x;
      m1[o + 1] = y;
      m1[o + 2] = z;
    }
  }
}

// ---- fused gather-and-fold (no loss): the chain walk of k_tracks_fill, but instead of writing the two
// rays of a residual block (48 B) for k_fold to read back (48 B), the block's c = (R_a m0) x (R_{a+s} m1)
// is formed on the spot and c c^T summed per (frame a, feature group, step): 6 sums per step and
// wavefront (wave_reduce8), kept in LDS for the segment's steps; at the end the workgroup's four wavefronts
// (four consecutive feature groups) are added up in group order and parked in `part`; k_tracks_reduce adds
// the workgroups up in order into the solver's per-pair sums, which is all k_lm reads without a loss.  The
// residual blocks never exist in HBM.
// part[((a * n_groups4 + g4) * SP + s - 1) * 8 + v]
// grid (groups / 4, M, segments): a step's partial sum is formed by the same lanes in the same order
// whichever wavefront takes the step.
#define TR_FOLD_WAVES 4  // wavefronts (64-feature groups) per k_tracks_fold workgroup: one stored partial for the four
__global__ __launch_bounds__(64 * TR_FOLD_WAVES) void k_tracks_fold(TracksDev t, uint32_t N, uint32_t n_groups4,
                                                                   uint32_t SP, uint32_t seg,
                                                                   const double* __restrict__ frames,
                                                                   double* __restrict__ part,
                                                                   double* __restrict__ pair_sums) {
  extern __shared__ double lp[];  // [TR_FOLD_WAVES][seg][8]: the wavefronts' partial sums of this segment's steps
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t g = blockIdx.x * TR_FOLD_WAVES + wave, a = blockIdx.y;
  const uint32_t slot_a = tr_slot(t, a), n = t.n_feat[slot_a];
  const uint32_t i = g * 64 + lane;
  const bool have = i < n;
  const uint32_t s_lo = blockIdx.z * seg;
  const uint32_t s_hi = min(s_lo + seg, N - 1 - a);  // the pairs (a, a + s - 1) exist for s <= N - 1 - a
  double* __restrict__ mine = lp + (size_t)wave * seg * 8;
  // k_tracks_reduce adds up EVERY group's entry of a step: a step no feature of these groups reaches is a
  // stored zero, not a hole (no count pass, no host in front of the fold)
  for (uint32_t e = lane; e < seg * 8; e += 64) mine[e] = 0.0;
  uint32_t c = have ? min(t.counter[(size_t)slot_a * t.max_features + i], N - 1 - a) : 0u;
  if (c <= s_lo) c = 0;  // nothing to do in this segment
  if (__ballot(c != 0)) {  // wave-uniform
    int32_t cur = (int32_t)i;
    if (s_lo && c) {
      cur = tr_seek(t, a, blockIdx.z, i);
      if (cur < 0) c = 0;
    }
    double px = 0.0, py = 0.0, pz = 0.0;  // R_a m0 (IterativeRefinement.cpp:172)
    if (c) {
      const double* __restrict__ r0 = t.rays + ((size_t)slot_a * t.max_features + i) * 3;
      const double* __restrict__ R0 = frames + (size_t)MVO_FRAME_DOUBLES * a;
      const double x = r0[0], y = r0[1], z = r0[2];
      px = R0[0] * x + R0[1] * y + R0[2] * z;
      py = R0[3] * x + R0[4] * y + R0[5] * z;
      pz = R0[6] * x + R0[7] * y + R0[8] * z;
    }
    uint32_t idx = (uint32_t)cur;
    uint32_t slot_s = tr_slot(t, a + s_lo);  // wave-uniform: the keyframe the walk stands in
    int32_t nxt = c ? t.pre[(size_t)slot_s * t.max_features + idx] : -1;
    for (uint32_t s = s_lo + 1; s <= s_hi; ++s) {
      const bool alive = c >= s;
      if (!__ballot(alive)) break;
      slot_s = tr_older(t, slot_s);
      double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (alive) {
        idx = min((uint32_t)nxt, t.max_features - 1);  // see k_tracks_fill
        const size_t cell = (size_t)slot_s * t.max_features + idx;
        const double* __restrict__ r1 = t.rays + cell * 3;
        const double x = r1[0], y = r1[1], z = r1[2];
        if (c > s) nxt = t.pre[cell];  // in flight together with the ray
        const double* __restrict__ R1 = frames + (size_t)MVO_FRAME_DOUBLES * (a + s);  // frames[b + 1], IR.cpp:171
        const double qx = R1[0] * x + R1[1] * y + R1[2] * z;
        const double qy = R1[3] * x + R1[4] * y + R1[5] * z;
        const double qz = R1[6] * x + R1[7] * y + R1[8] * z;
        const double vx = py * qz - pz * qy, vy = pz * qx - px * qz, vz = px * qy - py * qx;
        S[0] = vx * vx;
        S[1] = vx * vy;
        S[2] = vx * vz;
        S[3] = vy * vy;
        S[4] = vy * vz;
        S[5] = vz * vz;
      }
      const double tot = wave_reduce8(S, (int)lane);
      if ((lane & 7) == 0) mine[(size_t)(s - 1 - s_lo) * 8 + reduce8_value_of_lane((int)lane)] = tot;
    }
  }
  __syncthreads();
  // the four wavefronts' partial sums of a step, added in wavefront (= feature group) order: one entry
  // per workgroup and step
  // pair_sums != nullptr: ONE workgroup holds all the features of a frame (n_groups4 == 1), its sums are
  // the pairs' sums -- straight into the solver's table, no k_tracks_reduce launch
  double* __restrict__ prow = pair_sums ? pair_sums + (size_t)tr_pair_start(N, a) * 8
                                        : part + (size_t)(a * n_groups4 + blockIdx.x) * SP * 8;
  for (uint32_t e = threadIdx.x; e < seg * 8; e += 64 * TR_FOLD_WAVES) {
    const uint32_t s = s_lo + 1 + (e >> 3);
    if (s > s_hi) break;
    double sum = lp[e];
#pragma unroll
    for (int w = 1; w < TR_FOLD_WAVES; ++w) sum += lp[(size_t)w * seg * 8 + e];
    prow[(size_t)(s - 1) * 8 + (e & 7)] = sum;  // entries 6 and 7 are sums of zeros
  }
}

// pair (a, s): sum of the workgroups' partial sums in group order (k_tracks_fold stored one for every
// four feature groups of the frame, zeros included).  grid (N - 1, M), 256 threads = 32 group lanes x 8 values.
__global__ __launch_bounds__(256) void k_tracks_reduce(TracksDev t, uint32_t N, uint32_t n_groups4, uint32_t SP,
                                                      const double* __restrict__ part, double* __restrict__ pair_sums) {
  __shared__ double red[32][8];
  const uint32_t s1 = blockIdx.x, a = blockIdx.y;  // s - 1
  if (s1 >= N - 1 - a) return;
  const uint32_t v = threadIdx.x & 7, gl = threadIdx.x >> 3;
  const uint32_t ng = ((t.n_feat[tr_slot(t, a)] + 63) / 64 + TR_FOLD_WAVES - 1) / TR_FOLD_WAVES;
  const uint32_t p = tr_pair_start(N, a) + s1;
  double acc = 0.0;
  for (uint32_t g = gl; g < ng; g += 32) acc += part[((size_t)(a * n_groups4 + g) * SP + s1) * 8 + v];
  red[gl][v] = acc;
  __syncthreads();
  if (gl == 0) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += red[k][v];
    pair_sums[(size_t)p * 8 + v] = v < 6 ? sum : 0.0;
  }
}

// ---- the scale stage's three-frame gather (Scaler.cpp:22 -> Frame.hpp:103-152, depth 2): features of
// the frame `age` pushes ago whose track reaches back two more frames, in that frame's order.
// cnt[g] = such features in the 64-feature group g; one workgroup scans the groups (-> base ranks in
// place, total behind them); the fill follows the two predecessor links.
__global__ __launch_bounds__(64) void k_tracks_count3(TracksDev t, uint32_t age, uint32_t* __restrict__ cnt) {
  const uint32_t g = blockIdx.x, lane = threadIdx.x, slot = tr_slot(t, age), n = t.n_feat[slot];
  const uint32_t i = g * 64 + lane;
  const bool take = i < n && t.counter[(size_t)slot * t.max_features + i] >= 2;
  const unsigned long long bal = __ballot(take);
  if (lane == 0) cnt[g] = (uint32_t)__popcll(bal);
}

__global__ __launch_bounds__(1024) void k_tracks_scan3(uint32_t groups, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x, per = (groups + 1023) / 1024, lo = min(tid * per, groups), hi = min(lo + per, groups);
  uint32_t sum = 0;
  for (uint32_t g = lo; g < hi; ++g) sum += cnt[g];
  part[tid] = sum;
  __syncthreads();
  for (uint32_t o = 1; o < 1024; o <<= 1) {  // inclusive scan of the 1024 partial sums
    const uint32_t v = tid >= o ? part[tid - o] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;
  for (uint32_t g = lo; g < hi; ++g) {
    const uint32_t v = cnt[g];
    cnt[g] = run;
    run += v;
  }
  if (tid == 1023) cnt[groups] = part[1023];
}

__global__ __launch_bounds__(64) void k_tracks_fill3(TracksDev t, uint32_t age, const uint32_t* __restrict__ base,
                                                    double* __restrict__ v0, double* __restrict__ v1,
                                                    double* __restrict__ v2) {
  const uint32_t g = blockIdx.x, lane = threadIdx.x, slot0 = tr_slot(t, age), n = t.n_feat[slot0];
  const uint32_t i = g * 64 + lane;
  const bool take = i < n && t.counter[(size_t)slot0 * t.max_features + i] >= 2;
  const unsigned long long bal = __ballot(take);
  if (!take) return;
  const size_t o = 3 * (size_t)(base[g] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)));
  const uint32_t slot1 = tr_slot(t, age + 1), slot2 = tr_slot(t, age + 2);
  const uint32_t i1 = (uint32_t)t.pre[(size_t)slot0 * t.max_features + i];
  const uint32_t i2 = (uint32_t)t.pre[(size_t)slot1 * t.max_features + i1];
  const double* __restrict__ r0 = t.rays + ((size_t)slot0 * t.max_features + i) * 3;
  const double* __restrict__ r1 = t.rays + ((size_t)slot1 * t.max_features + i1) * 3;
  const double* __restrict__ r2 = t.rays + ((size_t)slot2 * t.max_features + i2) * 3;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    v0[o + q] = r0[q];
    v1[o + q] = r1[q];
    v2[o + q] = r2[q];
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
  int32_t* d_skip = nullptr;  // [max_frames][TR_SKIP_LEVELS][max_features]
  uint32_t *d_counter = nullptr, *d_nfeat = nullptr, *d_counts = nullptr;
  uint32_t* d_cnt = nullptr;  // [M][n_groups][SP]: per-group counts, then base ranks
  size_t cap_cnt = 0;
  double* d_part = nullptr;   // fused fold: [M][n_groups][SP][8] partial moments
  size_t cap_part = 0;
  double* d_frames = nullptr;      // fused fold: the window's frame records (it starts before the solver's upload)
  double* h_frames_pin = nullptr;  // pinned staging for them
  bool fold_wrote_sums = false;    // the last fold wrote the pairs' sums itself (one workgroup per frame)
  hipEvent_t ev_counts = nullptr;  // the pair counts have reached h_counts
  // the fused refine counts nothing; residual_blocks of mvo_hip_tracks_get_timing is counted when asked for
  uint64_t version = 0;            // bumped by push / append / reset
  uint64_t counted_version = ~0ull;
  uint32_t counted_M = 0, counted_N = 0, last_M = 0, last_N = 0;
  double* d_rays_in = nullptr;  // staging of the caller's rays
  uint32_t g_groups = 0, g_sp = 0;  // shape of d_cnt at the last count
  uint32_t* h_total = nullptr;      // pinned: number of three-frame tracks
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
  uint64_t last_blocks = 0, last_fill_bytes = 0, last_feature_bytes = 0;
  bool last_fused = false;
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

// walkers: TR_SEG steps per wavefront from a skip-link start; MVO_HIP_TRACKS_NO_SKIP (experiments, and the
// test that both give the same bytes): one wavefront walks the whole chain
void walk_shape(uint32_t N, uint32_t* seg, uint32_t* n_seg) {
  if (std::getenv("MVO_HIP_TRACKS_NO_SKIP")) {
    *seg = N - 1;
    *n_seg = 1;
  } else {
    *seg = TR_SEG;
    *n_seg = (N - 1 + TR_SEG - 1) / TR_SEG;
  }
}

TracksDev dev_view(const mvo_hip_tracks_context* t) {
  TracksDev d;
  d.rays = t->d_rays;
  d.pre = t->d_pre;
  d.skip = t->d_skip;
  d.counter = t->d_counter;
  d.n_feat = t->d_nfeat;
  d.max_frames = t->max_frames;
  d.max_features = t->max_features;
  d.head = t->head;
  return d;
}

// k_tracks_count / k_tracks_scan and the copy of the P pair counts to the host, queued; count_finish waits
// for that copy alone (an event, not the stream: the fused fold is queued behind it and keeps running
// while the host builds the solver's tables)
// argument checks of a window on the table, the pair buffers and the shape of the per-group tables
int window_shape(mvo_hip_tracks_context* t, uint32_t M, uint32_t N) {
  if (M == 0 || N <= M) return MVO_HIP_ERR_INVALID_ARGUMENT;            // IterativeRefinement.cpp:29-30
  if (M > MVO_HIP_MAX_REFINE || N > MVO_HIP_MAX_FRAMES) return MVO_HIP_ERR_UNSUPPORTED;
  if (N > t->n_pushed || N > t->max_frames) return MVO_HIP_ERR_STATE;  // not enough keyframes in the table
  TR_TRY(hipSetDevice(t->device));
  const uint32_t P = num_pairs(M, N);
  if (P > t->cap_pairs) {
    if (t->d_counts) (void)mvo_device_free(t->d_counts);
    if (t->d_offsets) (void)mvo_device_free(t->d_offsets);
    if (t->h_counts) (void)hipHostFree(t->h_counts);
    if (t->h_offsets) (void)hipHostFree(t->h_offsets);
    t->d_counts = nullptr;
    t->d_offsets = nullptr;
    t->h_counts = nullptr;
    t->h_offsets = nullptr;
    t->cap_pairs = 0;
    TR_TRY(mvo_device_alloc((void**)&t->d_counts, sizeof(uint32_t) * (P + 1)));
    TR_TRY(mvo_device_alloc((void**)&t->d_offsets, sizeof(uint64_t) * (P + 1)));
    TR_TRY(hipHostMalloc((void**)&t->h_counts, sizeof(uint32_t) * (P + 1), hipHostMallocDefault));
    TR_TRY(hipHostMalloc((void**)&t->h_offsets, sizeof(uint64_t) * (P + 1), hipHostMallocDefault));
    t->cap_pairs = P;
  }
  // feature groups of the widest of the M newest frames; the per-group tables hold [M][groups][SP] entries
  uint32_t groups = 0;
  for (uint32_t a = 0; a < M; ++a)
    groups = std::max(groups, (t->h_nfeat[(t->head + t->max_frames - a) % t->max_frames] + 63) / 64);
  t->g_groups = groups;
  t->g_sp = (N - 1 + 63) / 64 * 64;
  return MVO_HIP_OK;
}

int ensure_events(mvo_hip_tracks_context* t) {
  if (t->profiling && !t->events_ready) {
    for (auto& e : t->ev) TR_TRY(hipEventCreate(&e));
    t->events_ready = true;
  }
  return MVO_HIP_OK;
}

int count_launch(mvo_hip_tracks_context* t, uint32_t M, uint32_t N) {
  const int rc_shape = window_shape(t, M, N);
  if (rc_shape != MVO_HIP_OK) return rc_shape;
  const uint32_t P = num_pairs(M, N);
  hipStream_t s = t->stream;
  t->count_timed = false;
  const int rc_ev = ensure_events(t);
  if (rc_ev != MVO_HIP_OK) return rc_ev;
  const uint32_t groups = t->g_groups, SP = t->g_sp;
  const size_t cnt_elems = (size_t)M * groups * SP;
  if (cnt_elems > t->cap_cnt) {
    if (t->d_cnt) (void)mvo_device_free(t->d_cnt);
    t->d_cnt = nullptr;
    t->cap_cnt = 0;
    TR_TRY(mvo_device_alloc((void**)&t->d_cnt, sizeof(uint32_t) * cnt_elems));
    t->cap_cnt = cnt_elems;
  }
  if (t->profiling) TR_TRY(hipEventRecord(t->ev[0], s));
  if (groups) {
    hipLaunchKernelGGL(k_tracks_count, dim3(groups, M), dim3(64), 0, s, dev_view(t), N, groups, SP, t->d_cnt);
    TR_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_tracks_scan, dim3(SP / 64, M), dim3(64 * TR_SCAN_GROUPS), 0, s, dev_view(t), N, groups, SP,
                       t->d_cnt, t->d_counts);
    TR_TRY(hipGetLastError());
  } else {
    TR_TRY(hipMemsetAsync(t->d_counts, 0, sizeof(uint32_t) * P, s));
  }
  if (t->profiling) {
    TR_TRY(hipEventRecord(t->ev[1], s));
    t->count_timed = true;
  }
  TR_TRY(hipMemcpyAsync(t->h_counts, t->d_counts, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, s));
  if (!t->ev_counts) TR_TRY(hipEventCreateWithFlags(&t->ev_counts, hipEventDisableTiming));
  TR_TRY(hipEventRecord(t->ev_counts, s));
  return MVO_HIP_OK;
}

// host prefix sum: h_offsets / d_offsets hold the CSR offsets of the P pairs
int count_finish(mvo_hip_tracks_context* t, uint32_t M, uint32_t N) {
  const uint32_t P = num_pairs(M, N);
  hipStream_t s = t->stream;
  TR_TRY(hipEventSynchronize(t->ev_counts));
  t->h_offsets[0] = 0;
  for (uint32_t p = 0; p < P; ++p) t->h_offsets[p + 1] = t->h_offsets[p] + t->h_counts[p];
  {  // algorithmic bytes of the fill that follows (include/mvo_hip_tracks.h)
    uint64_t bytes = 0;
    for (uint32_t a = 0; a < M; ++a) bytes += 32ull * t->h_nfeat[(t->head + t->max_frames - a) % t->max_frames];
    t->last_blocks = t->h_offsets[P];
    t->last_feature_bytes = bytes;
    t->last_fill_bytes = bytes + 76ull * t->last_blocks;
  }
  t->counted_version = t->version;
  t->counted_M = M;
  t->counted_N = N;
  TR_TRY(hipMemcpyAsync(t->d_offsets, t->h_offsets, sizeof(uint64_t) * (P + 1), hipMemcpyHostToDevice, s));
  return MVO_HIP_OK;
}

int count_pairs(mvo_hip_tracks_context* t, uint32_t M, uint32_t N) {
  const int rc = count_launch(t, M, N);
  return rc != MVO_HIP_OK ? rc : count_finish(t, M, N);
}

// k_tracks_fill into m0 / m1 (device, room for h_offsets[P] blocks); queued on the stream
int fill_pairs(mvo_hip_tracks_context* t, uint32_t M, uint32_t N, double* m0, double* m1) {
  const uint32_t P = num_pairs(M, N);
  t->last_fused = false;
  t->fill_timed = false;
  if (t->h_offsets[P] == 0) return MVO_HIP_OK;
  if (t->profiling) TR_TRY(hipEventRecord(t->ev[2], t->stream));
  uint32_t seg = 0, n_seg = 0;
  walk_shape(N, &seg, &n_seg);
  hipLaunchKernelGGL(k_tracks_fill, dim3(t->g_groups, M, n_seg), dim3(64), 0, t->stream, dev_view(t), N, t->g_groups,
                     t->g_sp, seg, t->d_cnt, t->d_offsets, m0, m1);
  TR_TRY(hipGetLastError());
  if (t->profiling) {
    TR_TRY(hipEventRecord(t->ev[3], t->stream));
    t->fill_timed = true;
  }
  return MVO_HIP_OK;
}

// fused gather-and-fold (no loss), first half: k_tracks_fold, per (frame a, feature group, step) partial
// moments straight from the track tables into d_part.  Needs the tables, the frame records on the device
// and the shape window_shape left (g_groups, g_sp) -- no counts, nothing from the host.
int fold_launch(mvo_hip_tracks_context* t, uint32_t M, uint32_t N, const double* frames, double* pair_sums = nullptr) {
  t->last_fused = true;
  t->fill_timed = false;
  t->fold_wrote_sums = false;
  hipStream_t s = t->stream;
  if (t->g_groups == 0) return MVO_HIP_OK;
  const int rc_ev = ensure_events(t);
  if (rc_ev != MVO_HIP_OK) return rc_ev;
  const uint32_t groups4 = (t->g_groups + TR_FOLD_WAVES - 1) / TR_FOLD_WAVES;
  const size_t part_elems = (size_t)M * groups4 * t->g_sp * 8;
  if (part_elems > t->cap_part) {
    if (t->d_part) (void)mvo_device_free(t->d_part);
    t->d_part = nullptr;
    t->cap_part = 0;
    TR_TRY(mvo_device_alloc((void**)&t->d_part, sizeof(double) * part_elems));
    t->cap_part = part_elems;
  }
  if (t->profiling) TR_TRY(hipEventRecord(t->ev[2], s));
  uint32_t seg = 0, n_seg = 0;
  walk_shape(N, &seg, &n_seg);
  hipLaunchKernelGGL(k_tracks_fold, dim3(groups4, M, n_seg), dim3(64 * TR_FOLD_WAVES),
                     sizeof(double) * 8 * TR_FOLD_WAVES * seg, s, dev_view(t), N, groups4, t->g_sp, seg, frames,
                     t->d_part, groups4 == 1 ? pair_sums : (double*)nullptr);
  t->fold_wrote_sums = groups4 == 1 && pair_sums != nullptr;
  TR_TRY(hipGetLastError());
  if (t->profiling) {
    TR_TRY(hipEventRecord(t->ev[3], s));
    t->fill_timed = true;
  }
  return MVO_HIP_OK;
}

// second half: k_tracks_reduce adds the groups up into pair_sums (device, 8 doubles per pair), on the
// stream behind fold_launch
int reduce_launch(mvo_hip_tracks_context* t, uint32_t M, uint32_t N, double* pair_sums) {
  const uint32_t P = num_pairs(M, N);
  hipStream_t s = t->stream;
  if (t->g_groups == 0) {
    TR_TRY(hipMemsetAsync(pair_sums, 0, sizeof(double) * 8 * P, s));
    return MVO_HIP_OK;
  }
  if (t->fold_wrote_sums) return MVO_HIP_OK;  // one workgroup per frame: the fold's sums are the pairs' sums
  hipLaunchKernelGGL(k_tracks_reduce, dim3(N - 1, M), dim3(256), 0, s, dev_view(t), N,
                     (t->g_groups + TR_FOLD_WAVES - 1) / TR_FOLD_WAVES, t->g_sp, t->d_part, pair_sums);
  TR_TRY(hipGetLastError());
  return MVO_HIP_OK;
}

// the window's frame records on the device, through the pinned staging block
int stage_frames(mvo_hip_tracks_context* t, const mvo_hip_frame* frames, uint32_t N) {
  static_assert(sizeof(mvo_hip_frame) == sizeof(double) * MVO_FRAME_DOUBLES, "frame record layout");
  if (!t->d_frames) TR_TRY(mvo_device_alloc((void**)&t->d_frames, sizeof(mvo_hip_frame) * MVO_HIP_MAX_FRAMES));
  if (!t->h_frames_pin)
    TR_TRY(hipHostMalloc((void**)&t->h_frames_pin, sizeof(mvo_hip_frame) * MVO_HIP_MAX_FRAMES, hipHostMallocDefault));
  std::memcpy(t->h_frames_pin, frames, sizeof(mvo_hip_frame) * N);
  TR_TRY(hipMemcpyAsync(t->d_frames, t->h_frames_pin, sizeof(mvo_hip_frame) * N, hipMemcpyHostToDevice, t->stream));
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
                  mvo_device_alloc((void**)&t->d_rays, sizeof(double) * 3 * cells) == hipSuccess &&
                  mvo_device_alloc((void**)&t->d_pre, sizeof(int32_t) * cells) == hipSuccess &&
                  mvo_device_alloc((void**)&t->d_skip, sizeof(int32_t) * TR_SKIP_LEVELS * cells) == hipSuccess &&
                  mvo_device_alloc((void**)&t->d_counter, sizeof(uint32_t) * cells) == hipSuccess &&
                  mvo_device_alloc((void**)&t->d_pre_in, sizeof(int32_t) * max_features) == hipSuccess &&
                  mvo_device_alloc((void**)&t->d_rays_in, sizeof(double) * 3 * max_features) == hipSuccess &&
                  mvo_device_alloc((void**)&t->d_nfeat, sizeof(uint32_t) * max_frames) == hipSuccess &&
                  // on the table's stream: the null stream's memset may still be pending when the first push
                  // (a non-blocking stream, which does not wait for it) writes its feature count
                  hipMemsetAsync(t->d_nfeat, 0, sizeof(uint32_t) * max_frames, t->stream) == hipSuccess;
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
  void* bufs[] = {t->d_rays, t->d_pre, t->d_skip, t->d_counter, t->d_nfeat, t->d_counts, t->d_pre_in, t->d_offsets, t->d_m0, t->d_m1,
                  t->d_cnt, t->d_rays_in, t->d_part, t->d_frames};
  for (void* p : bufs)
    if (p) (void)mvo_device_free(p);
  if (t->h_counts) (void)hipHostFree(t->h_counts);
  if (t->h_offsets) (void)hipHostFree(t->h_offsets);
  if (t->h_total) (void)hipHostFree(t->h_total);
  if (t->h_frames_pin) (void)hipHostFree(t->h_frames_pin);
  if (t->ev_counts) (void)hipEventDestroy(t->ev_counts);
  if (t->events_ready)
    for (auto& e : t->ev) (void)hipEventDestroy(e);
  delete t;
}

int mvo_hip_tracks_reset(mvo_hip_tracks_handle t) {
  if (!t) return MVO_HIP_ERR_INVALID_ARGUMENT;
  TR_TRY(hipSetDevice(t->device));
  TR_TRY(hipStreamSynchronize(t->stream));
  t->n_pushed = 0;
  ++t->version;
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
    TR_TRY(hipMemcpyAsync(t->d_rays_in, rays, sizeof(double) * 3 * n_features, hipMemcpyHostToDevice, s));
    TR_TRY(hipMemcpyAsync(t->d_pre_in, pre_feature, sizeof(int32_t) * n_features, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_tracks_counter, dim3((n_features + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, s,
                       n_features, t->d_rays_in, t->d_pre_in, t->d_counter + (size_t)prev * t->max_features, prev_n,
                       t->d_rays + (size_t)slot * t->max_features * 3, t->d_pre + (size_t)slot * t->max_features,
                       t->d_counter + (size_t)slot * t->max_features);
    TR_TRY(hipGetLastError());
    TracksDev d = dev_view(t);
    d.head = slot;  // the frame being pushed is age 0 for its skip links
    const uint32_t n_avail = std::min<uint64_t>((uint64_t)t->n_pushed + 1, t->max_frames);
    hipLaunchKernelGGL(k_tracks_skip, dim3((n_features + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, s, d, 0u,
                       n_features, n_avail, t->d_skip);
    TR_TRY(hipGetLastError());
  }
  TR_TRY(hipMemcpyAsync(t->d_nfeat + slot, &n_features, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  TR_TRY(hipStreamSynchronize(s));  // the caller's arrays (and n_features on our stack) may go away
  t->h_nfeat[slot] = n_features;
  t->head = slot;
  if (t->n_pushed < 0xFFFFFFFFu) ++t->n_pushed;
  ++t->version;
  return MVO_HIP_OK;
}

int mvo_hip_tracks_append(mvo_hip_tracks_handle t, uint32_t n_features, const double* rays) {
  if (!t || (n_features && !rays)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (t->n_pushed == 0) return MVO_HIP_ERR_STATE;
  const uint32_t slot = t->head, n_old = t->h_nfeat[slot];
  if (n_features > t->max_features - n_old) return MVO_HIP_ERR_UNSUPPORTED;
  if (n_features == 0) return MVO_HIP_OK;
  TR_TRY(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  const uint32_t n_new = n_old + n_features;
  const size_t cell = (size_t)slot * t->max_features + n_old;
  TR_TRY(hipMemcpyAsync(t->d_rays_in, rays, sizeof(double) * 3 * n_features, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_tracks_counter, dim3((n_features + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, s,
                     n_features, t->d_rays_in, (const int32_t*)nullptr, (const uint32_t*)nullptr, 0u,
                     t->d_rays + cell * 3, t->d_pre + cell, t->d_counter + cell);
  TR_TRY(hipGetLastError());
  // new tracks: counter 0, so every skip link comes out -1
  hipLaunchKernelGGL(k_tracks_skip, dim3((n_features + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, s, dev_view(t),
                     n_old, n_new, (uint32_t)std::min<uint64_t>(t->n_pushed, t->max_frames), t->d_skip);
  TR_TRY(hipGetLastError());
  TR_TRY(hipMemcpyAsync(t->d_nfeat + slot, &n_new, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  TR_TRY(hipStreamSynchronize(s));
  t->h_nfeat[slot] = n_new;
  ++t->version;
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
    if (t->d_m0) (void)mvo_device_free(t->d_m0);
    if (t->d_m1) (void)mvo_device_free(t->d_m1);
    t->d_m0 = t->d_m1 = nullptr;
    t->cap_obs = 0;
    const uint64_t want = total + total / 4 + 64;
    TR_TRY(mvo_device_alloc((void**)&t->d_m0, sizeof(double) * 3 * want));
    TR_TRY(mvo_device_alloc((void**)&t->d_m1, sizeof(double) * 3 * want));
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
  // Without a loss the LM loop reads nothing but 8 sums per pair: they are formed straight from the
  // tables (k_tracks_fold) and the residual blocks never reach HBM -- nor are they counted: nothing on
  // this path waits for the device before the solve is queued.  With a loss (or per-evaluation passes)
  // the blocks are counted, their offsets summed up on the host, and gathered into the solver's ray
  // buffers as before.
  const bool fused = !options->use_loss_function && !(options->assembly_mode & MVO_HIP_ASSEMBLY_PASSES) &&
                     !std::getenv("MVO_HIP_TRACKS_NO_FUSE");
  static double placeholder[3] = {0.0, 0.0, 1.0};  // never read: the rays are written on the device
  mvo_hip_observations obs;
  obs.n_pairs = num_pairs(M, N);
  obs.reserved = 0;
  obs.m0 = placeholder;
  obs.m1 = placeholder;
  int rc;
  if (fused) {
    rc = window_shape(t, M, N);
    if (rc != MVO_HIP_OK) return rc;
    // A long fold (many walker wavefronts) is queued first, on a copy of the frame records of its own,
    // and runs while the host builds the solver's tables; a short one is not worth the second copy.
    const bool fold_first = (uint64_t)t->g_groups * M * ((N + TR_SEG - 2) / TR_SEG) > 256;
    if (fold_first) {
      rc = stage_frames(t, frames, N);
      if (rc != MVO_HIP_OK) return rc;
      rc = fold_launch(t, M, N, t->d_frames);
      if (rc != MVO_HIP_OK) return rc;
    }
    // the solver is told of no residual blocks: it builds no tile tables and reads the sums it is handed
    for (uint32_t p = 0; p <= obs.n_pairs; ++p) t->h_offsets[p] = 0;
    obs.pair_offsets = t->h_offsets;
    const double* frames_dev = nullptr;
    double* sums = nullptr;
    rc = mvo_hip_internal_upload_pair_sums(t->solver, frames, M, N, options, &obs, &frames_dev, &sums);
    if (rc != MVO_HIP_OK) return rc;
    if (!fold_first) {
      rc = fold_launch(t, M, N, frames_dev, sums);
      if (rc != MVO_HIP_OK) return rc;
    }
    rc = reduce_launch(t, M, N, sums);
    if (rc != MVO_HIP_OK) return rc;
    t->last_M = M;
    t->last_N = N;
  } else {
    rc = count_pairs(t, M, N);
    if (rc != MVO_HIP_OK) return rc;
    obs.pair_offsets = t->h_offsets;
    double *m0 = nullptr, *m1 = nullptr;
    rc = mvo_hip_internal_upload_device_rays(t->solver, frames, M, N, options, &obs, &m0, &m1);
    if (rc != MVO_HIP_OK) return rc;
    rc = fill_pairs(t, M, N, m0, m1);  // straight into the solver's ray buffers
    if (rc != MVO_HIP_OK) return rc;
  }
  rc = mvo_hip_solve(t->solver);
  if (rc != MVO_HIP_OK) return rc;
  return mvo_hip_download(t->solver, frames, summary);
}

int mvo_hip_tracks_pair_moments(mvo_hip_tracks_handle t, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                                double* moments) {
  if (!t || !frames || !moments) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = count_pairs(t, M, N);
  if (rc != MVO_HIP_OK) return rc;
  const uint32_t P = num_pairs(M, N);
  hipStream_t s = t->stream;
  double *d_frames = nullptr, *d_sums = nullptr;
  TR_TRY(mvo_device_alloc((void**)&d_frames, sizeof(mvo_hip_frame) * N));
  hipError_t e = mvo_device_alloc((void**)&d_sums, sizeof(double) * 8 * P);
  if (e == hipSuccess) e = hipMemcpyAsync(d_frames, frames, sizeof(mvo_hip_frame) * N, hipMemcpyHostToDevice, s);
  std::vector<double> sums(8 * (size_t)P);
  if (e == hipSuccess) {
    rc = fold_launch(t, M, N, d_frames);
    if (rc == MVO_HIP_OK) rc = reduce_launch(t, M, N, d_sums);
    if (rc == MVO_HIP_OK) e = hipMemcpyAsync(sums.data(), d_sums, sizeof(double) * 8 * P, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)mvo_device_free(d_frames);
  if (d_sums) (void)mvo_device_free(d_sums);
  if (rc != MVO_HIP_OK) return rc;
  TR_TRY(e);
  for (uint32_t p = 0; p < P; ++p)
    for (int v = 0; v < 6; ++v) moments[(size_t)p * 6 + v] = sums[(size_t)p * 8 + v];
  return MVO_HIP_OK;
}

int mvo_hip_tracks_scale(mvo_hip_tracks_handle t, mvo_hip_scaler_handle scaler, uint32_t age,
                         const mvo_hip_scaler_input* in, double* per_track, mvo_hip_scaler_result* result) {
  if (!t || !scaler || !in || !result) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if ((uint64_t)age + 3 > t->n_pushed || (uint64_t)age + 3 > t->max_frames) return MVO_HIP_ERR_STATE;
  TR_TRY(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  const uint32_t n = t->h_nfeat[(t->head + t->max_frames - age) % t->max_frames];
  const uint32_t groups = (n + 63) / 64;
  if ((size_t)groups + 1 > t->cap_cnt) {
    if (t->d_cnt) (void)mvo_device_free(t->d_cnt);
    t->d_cnt = nullptr;
    t->cap_cnt = 0;
    TR_TRY(mvo_device_alloc((void**)&t->d_cnt, sizeof(uint32_t) * ((size_t)groups + 1)));
    t->cap_cnt = (size_t)groups + 1;
  }
  if (!t->h_total) TR_TRY(hipHostMalloc((void**)&t->h_total, sizeof(uint32_t) * 4, hipHostMallocDefault));
  t->h_total[0] = 0;
  const TracksDev d = dev_view(t);
  if (groups) {
    hipLaunchKernelGGL(k_tracks_count3, dim3(groups), dim3(64), 0, s, d, age, t->d_cnt);
    hipLaunchKernelGGL(k_tracks_scan3, dim3(1), dim3(1024), 0, s, groups, t->d_cnt);
    TR_TRY(hipGetLastError());
    TR_TRY(hipMemcpyAsync(t->h_total, t->d_cnt + groups, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    TR_TRY(hipStreamSynchronize(s));
  }
  const uint32_t total = t->h_total[0];
  double *v0 = nullptr, *v1 = nullptr, *v2 = nullptr;
  int sc_device = -1;
  int rc = mvo_hip_scaler_internal_reserve(scaler, total, &v0, &v1, &v2, &sc_device);
  if (rc != MVO_HIP_OK) return rc;
  if (sc_device != t->device) return MVO_HIP_ERR_INVALID_ARGUMENT;  // both handles must live on one GPU
  TR_TRY(hipSetDevice(t->device));
  if (total) {
    hipLaunchKernelGGL(k_tracks_fill3, dim3(groups), dim3(64), 0, s, d, age, t->d_cnt, v0, v1, v2);
    TR_TRY(hipGetLastError());
    TR_TRY(hipStreamSynchronize(s));  // the scale stage runs on its own stream
  }
  return mvo_hip_scaler_run(scaler, in, per_track, result);
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
  if (t->last_fused && t->last_N != 0 &&
      (t->counted_version != t->version || t->counted_M != t->last_M || t->counted_N != t->last_N)) {
    // the fused refine did not count its residual blocks; the caller wants to know: count them now
    const bool prof = t->profiling;
    t->profiling = false;  // not part of the refine that was timed
    const int rc = count_pairs(t, t->last_M, t->last_N);
    t->profiling = prof;
    if (rc != MVO_HIP_OK) return rc;
    TR_TRY(hipStreamSynchronize(t->stream));
  }
  out->residual_blocks = t->last_blocks;
  out->fill_bytes = t->last_fused ? t->last_feature_bytes + 28ull * t->last_blocks  // per block: predecessor index + ray
                                  : t->last_fill_bytes;
  return MVO_HIP_OK;
}

}  // extern "C"

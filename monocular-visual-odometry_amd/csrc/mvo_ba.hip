// mvo_ba.hip -- explicit-landmark window bundle adjustment on gfx950: the C ABI of
// include/mvo_hip_ba.h.  No counterpart in the reference; the
// problem statement and solver rules are written down in include/mvo_hip_ba.h.  No CPU fallback:
// every entry point needs a HIP device.
//
// One LM iteration = five launches, enqueued without host round trips:
//   k_ba_schur    one WAVEFRONT per landmark: per-observation residual + analytic Jacobians,
//                 landmark block C_j (wave reduction), (C_j + D)^-1, Y = E C^-1 staged in LDS; then
//                 one THREAD per 6 x 6 block of the reduced camera system, the block in registers
//                 for the whole kernel, subtracts Y_s E_s'^T (no atomics, fixed order)
//   k_ba_reduce   sums the per-workgroup copies (fixed order)
//   k_ba_solve    one workgroup: damping, dense Cholesky of the <= 168 x 168 system in LDS with the
//                 right-hand side as an extra row, back substitution, candidate poses
//   k_ba_step     one wavefront per landmark: back substitution of the landmark step, candidate
//                 landmark, cost at the candidate
//   k_ba_decide   trust-region accept/reject on the device-resident state
#include "../../include/mvo_hip_ba.h"
#include "../../include/mvo_hip.h"

#include <hip/hip_runtime.h>

#include "mvo_alloc.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mvo_wave.h"

#define BA_THREADS 256
#define BA_WAVES 4
#define BA_MAXF MVO_HIP_BA_MAX_FREE_FRAMES
#define BA_SCHUR_THREADS 512
#define BA_SCHUR_WAVES 8
#define BA_STAGE 51                      // doubles per free observation in a wave's LDS stage:
                                         // E(18) Y(18) sqrt(w) J_c(12) sqrt(w) r(2) frame(1)
#define BA_SOLVE_THREADS 512
#ifndef BA_PREFETCH
#define BA_PREFETCH 4  // rounds of observations in flight per lane in k_ba_fixed / k_ba_step_g
#endif
#define BA_NOSLOT 0xFFFFu

struct BaParams {
  double ftol, gtol, ptol, loss_scale, initial_radius;
  int32_t max_it, loss, M, N;
  uint32_t L, n_blk, n_wg, n_wg_step, trace_cap, n_wg_fixed;  // n_wg_fixed: workgroups of k_ba_fixed (0: not launched)
};

struct BaState {
  double radius, decf, cost, initial_cost, gmax, mcc_c, step2_c, xnorm2_c;
  int32_t cur, iteration, status, termination, nsucc, nunsucc, num_invalid, step_bad, ntrace, pad;
};

struct BaTrace {  // == mvo_hip_ba_iteration
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int32_t step_is_valid, step_is_successful;
};
static_assert(sizeof(BaTrace) == sizeof(mvo_hip_ba_iteration), "trace record layout");
static_assert(sizeof(mvo_hip_ba_pose) == 96, "pose record layout");

// ---------------------------------------------------------------------------------------------
// device math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ba_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ void ba_loss(int loss, double a, double s, double& rho, double& w) {
  const double b = a * a;
  if (loss == MVO_HIP_BA_LOSS_CAUCHY) {
    rho = b * log(1.0 + s / b);
    w = 1.0 / (1.0 + s / b);
  } else if (loss == MVO_HIP_BA_LOSS_HUBER && s > b) {
    const double rt = sqrt(s);
    rho = 2.0 * a * rt - b;
    w = a / rt;
  } else {
    rho = s;
    w = 1.0;
  }
}

// q = R^T (X - p), residual against the normalised image point (u, v)
__device__ __forceinline__ void ba_project(const double* __restrict__ pose, const double X[3], double& a, double& b,
                                           double& c) {
  const double d0 = X[0] - pose[9], d1 = X[1] - pose[10], d2 = X[2] - pose[11];
  a = pose[0] * d0 + pose[3] * d1 + pose[6] * d2;
  b = pose[1] * d0 + pose[4] * d1 + pose[7] * d2;
  c = pose[2] * d0 + pose[5] * d1 + pose[8] * d2;
}

struct BaObs {
  double r0, r1, w, rho;
  double jc[12];  // 2 x 6: omega, p
  double jx[6];   // 2 x 3
};

__device__ __forceinline__ void ba_obs_eval(const double* __restrict__ pose, const double X[3], double u, double v,
                                            int loss, double ls, BaObs& e) {
  double a, b, c;
  ba_project(pose, X, a, b, c);
  const double ic = ba_rcp(c);
  const double ai = a * ic, bi = b * ic;
  e.r0 = ai - u;
  e.r1 = bi - v;
  ba_loss(loss, ls, e.r0 * e.r0 + e.r1 * e.r1, e.rho, e.w);
  // J_X = Pi R^T with Pi = [[1/c, 0, -a/c^2], [0, 1/c, -b/c^2]]
  const double pa = -ai * ic, pb = -bi * ic;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e.jx[k] = ic * pose[3 * k] + pa * pose[3 * k + 2];
    e.jx[3 + k] = ic * pose[3 * k + 1] + pb * pose[3 * k + 2];
    e.jc[3 + k] = -e.jx[k];  // J_p = -J_X
    e.jc[9 + k] = -e.jx[3 + k];
  }
  // J_omega = Pi [q]x for R <- R Exp(omega)
  e.jc[0] = ai * bi;
  e.jc[1] = -1.0 - ai * ai;
  e.jc[2] = bi;
  e.jc[6] = 1.0 + bi * bi;
  e.jc[7] = -ai * bi;
  e.jc[8] = -ai;
}

__device__ __forceinline__ double ba_clamp_diag(double v) { return fmin(fmax(v, 1e-6), 1e32); }

// inverse of a symmetric 3x3 given as (xx, xy, xz, yy, yz, zz); same packing out
__device__ __forceinline__ void ba_inv3(const double C[6], double Ci[6]) {
  const double co0 = C[3] * C[5] - C[4] * C[4], co1 = C[2] * C[4] - C[1] * C[5], co2 = C[1] * C[4] - C[2] * C[3];
  const double id = 1.0 / (C[0] * co0 + C[1] * co1 + C[2] * co2);
  Ci[0] = co0 * id;
  Ci[1] = co1 * id;
  Ci[2] = co2 * id;
  Ci[3] = (C[0] * C[5] - C[2] * C[2]) * id;
  Ci[4] = (C[1] * C[2] - C[0] * C[4]) * id;
  Ci[5] = (C[0] * C[3] - C[1] * C[1]) * id;
}

// upper block triangle of the reduced camera system, block-row-major, 36 doubles per block
__device__ __host__ __forceinline__ uint32_t ba_blk(uint32_t M, uint32_t f, uint32_t f2) {  // f <= f2
  return f * M - (f * (f - 1)) / 2 + (f2 - f);
}

// 64-lane butterflies in registers: v_permlane32/16_swap across the rows, DPP inside a row (no trip
// through the LDS crossbar); every lane ends with the same total
__device__ __forceinline__ double ba_wave_sum(double v) {
  v = swap_add32(v, v);
  v = swap_add16(v, v);
  v += dpp_mov_f64<0x128>(v);  // row_ror:8
  v += dpp_mov_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);  // row_half_mirror
  return v;
}
__device__ __forceinline__ double ba_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 8; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  v = fmax(v, dpp_mov_f64<0x128>(v));
  v = fmax(v, dpp_mov_f64<0xB1>(v));
  v = fmax(v, dpp_mov_f64<0x4E>(v));
  v = fmax(v, dpp_mov_f64<0x141>(v));
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_ba_schur: 8 wavefronts per workgroup, one workgroup per CU.
//   phase A  one WAVEFRONT per landmark (8 at a time): lanes walk the landmark's observations --
//            residual, analytic Jacobians, loss weight; C_j and g_l by a wave reduction;
//            (C_j + D)^-1; per free observation the blocks E, Y = E C^-1, sqrt(w) J_c and
//            sqrt(w) r are staged in the wave's LDS area
//   phase B  one THREAD per 6 x 6 block (f, f2), f <= f2, of the reduced camera system: the block
//            lives in 36 registers for the whole kernel; for each staged landmark that is seen in
//            both frames it subtracts Y_s E_s2^T; diagonal owners also add J_c^T J_c and keep
//            the frame's vectors (rhs, raw gradient, diag B)
// No atomics of any kind, no LDS read-modify-write, fixed summation order => bitwise reproducible.
// partS[g][n_blk * 36], partV[g][18 M + 2] = rhs(6M) | g_c(6M) | diag(B)(6M) | cost | gmax(landmarks)
// ---------------------------------------------------------------------------------------------
#ifdef MVO_BA_STAMPS  // diagnostic build: cycles of the phases of k_ba_solve ([0..5]) and k_ba_schur ([8..11]), thread 0
__device__ unsigned long long mvo_ba_stamps[16];
#define BA_STAMP(i)                                                   \
  do {                                                                \
    if (threadIdx.x == 0) {                                           \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
      ba_acc_[(i)] += now_ - ba_prev_;                                \
      ba_prev_ = now_;                                                \
    }                                                                 \
  } while (0)
#else
#define BA_STAMP(i) do { } while (0)
#endif

// global -> LDS copy of n doubles (n even, both 16-byte aligned) by all nt threads: 16-byte loads, four
// in flight per thread before the first store (a one-load-per-iteration loop pays the L2 latency per
// iteration)
__device__ __forceinline__ void ba_stage_lds(double* dst, const double* __restrict__ src, uint32_t n, uint32_t tid,
                                             uint32_t nt) {
  const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const uint32_t n2 = n / 2;
  uint32_t i = tid;
  for (; i + 3 * nt < n2; i += 4 * nt) {
    const double2 a = s2[i], b = s2[i + nt], c = s2[i + 2 * nt], d = s2[i + 3 * nt];
    d2[i] = a;
    d2[i + nt] = b;
    d2[i + 2 * nt] = c;
    d2[i + 3 * nt] = d;
  }
  for (; i < n2; i += nt) d2[i] = s2[i];
}

struct SchurHdr {  // per staged landmark
  double gl[3];
  int32_t n_free, pad;
  int8_t slot_of[BA_MAXF + 4];  // frame -> slot of the landmark's observation in it, -1 if none
};

struct LmPre {  // one landmark's header and first 64 observations, fetched one group ahead
  uint32_t lm, o0, T, Tf, word;
  double X[3];
  double2 uv;
};

template <bool POSES_IN_LDS>
__global__ __launch_bounds__(BA_SCHUR_THREADS) void k_ba_schur(BaParams prm, const BaState* __restrict__ st,
                                                              double radius_override,
                                                              const double* __restrict__ poses2,
                                                              const double* __restrict__ X2,
                                                              const uint32_t* __restrict__ lm_off,
                                                              const uint32_t* __restrict__ lm_nfree,
                                                              const uint32_t* __restrict__ obs_frame,
                                                              const double2* __restrict__ obs_uv,
                                                              const uint32_t* __restrict__ wg_lm,
                                                              const uint32_t* __restrict__ lm_list,
                                                              double* __restrict__ partS,
                                                              double* __restrict__ partV,
                                                              double* __restrict__ lmC) {
  extern __shared__ double lds_poses[];  // 12 N doubles when POSES_IN_LDS
  __shared__ double stage[BA_SCHUR_WAVES][BA_MAXF * BA_STAGE];
  __shared__ SchurHdr hdr[BA_SCHUR_WAVES];
  __shared__ double red[2 * BA_SCHUR_WAVES];
  if (st->status != 0) return;
  const int M = prm.M, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t nS = prm.n_blk * 36, nV = 18 * M;
  const int cur = st->cur;
  const double radius = radius_override > 0.0 ? radius_override : st->radius;
  const double inv_radius = 1.0 / radius;
  const double* __restrict__ poses_g = poses2 + (size_t)cur * 12 * prm.N;
  const double* __restrict__ Xs = X2 + (size_t)cur * 3 * prm.L;
  const uint32_t g = blockIdx.x;
  if (POSES_IN_LDS) {
    ba_stage_lds(lds_poses, poses_g, 12 * (uint32_t)prm.N, (uint32_t)tid, BA_SCHUR_THREADS);
    __syncthreads();
  }
  const double* poses = POSES_IN_LDS ? lds_poses : poses_g;
  // block owned by this thread: tid = ba_blk(M, bf, bf2)
  const bool owner = (uint32_t)tid < prm.n_blk;
  int bf = 0;
  for (int q = 1; q < M; ++q)
    if ((uint32_t)tid >= ba_blk(M, q, q)) bf = q;
  const int bf2 = bf + (tid - (int)ba_blk(M, bf, bf));
  const bool diag_owner = owner && bf == bf2;
  double acc[36], vr[6], vg[6], vb[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) vr[e] = vg[e] = vb[e] = 0.0;
  double* my = stage[wave];
  double cost_acc = 0.0, gmax_acc = 0.0;
  const uint32_t lm_begin = wg_lm[g], lm_end = wg_lm[g + 1];
  // software pipeline over the landmarks of this wavefront: the header of landmark i+2 and the
  // observations of landmark i+1 are in flight while landmark i is processed
  // wg_lm cuts POSITIONS of lm_list (the landmarks with an observation in a free frame; those
  // without go through k_ba_fixed) or, without a list, landmark indices
  auto fetch_hdr = [&](uint32_t pos, LmPre& p) {
    p.lm = p.o0 = p.T = p.Tf = 0;
    if (pos < lm_end) {
      const uint32_t lm = lm_list ? lm_list[pos] : pos;
      p.lm = lm;
      p.o0 = lm_off[lm];
      p.T = lm_off[lm + 1] - p.o0;
      p.Tf = lm_nfree[lm];
      p.X[0] = Xs[3 * lm];
      p.X[1] = Xs[3 * lm + 1];
      p.X[2] = Xs[3 * lm + 2];
    }
  };
  auto fetch_obs = [&](LmPre& p) {
    if ((uint32_t)lane < p.T) {
      p.word = obs_frame[p.o0 + lane];
      p.uv = obs_uv[p.o0 + lane];
    }
  };
#ifdef MVO_BA_STAMPS
  unsigned long long ba_acc_[16] = {0}, ba_prev_ = __builtin_amdgcn_s_memtime();
#endif
  LmPre pa, pb;
  fetch_hdr(lm_begin + wave, pa);
  fetch_obs(pa);
  fetch_hdr(lm_begin + wave + BA_SCHUR_WAVES, pb);
  for (uint32_t lm0 = lm_begin; lm0 < lm_end; lm0 += BA_SCHUR_WAVES) {
    // ---------------- phase A ----------------
    const uint32_t pos = lm0 + wave;
    const LmPre pc = pa;
    fetch_obs(pb);
    pa = pb;
    fetch_hdr(pos + 2 * BA_SCHUR_WAVES, pb);
    const uint32_t Tf = pc.Tf, lm = pc.lm;
    if (pos < lm_end) {
      const uint32_t o0 = pc.o0, T = pc.T;
      const double X[3] = {pc.X[0], pc.X[1], pc.X[2]};
      if (Tf && lane < BA_MAXF + 4) hdr[wave].slot_of[lane] = -1;
      __builtin_amdgcn_wave_barrier();
      double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a9 = 0.0;  // C(6) gl(2) | gl[2]
      for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t t = base + lane;
        if (t < T) {
          uint32_t word = pc.word;
          double2 uv = pc.uv;
          if (base) {
            word = obs_frame[o0 + t];
            uv = obs_uv[o0 + t];
          }
          const uint32_t f = word & 0xFFFFu, slot = word >> 16;
          BaObs e;
          ba_obs_eval(poses + 12 * f, X, uv.x, uv.y, prm.loss, prm.loss_scale, e);
          cost_acc += 0.5 * e.rho;
          const double w = e.w;
          a8[0] += w * (e.jx[0] * e.jx[0] + e.jx[3] * e.jx[3]);
          a8[1] += w * (e.jx[0] * e.jx[1] + e.jx[3] * e.jx[4]);
          a8[2] += w * (e.jx[0] * e.jx[2] + e.jx[3] * e.jx[5]);
          a8[3] += w * (e.jx[1] * e.jx[1] + e.jx[4] * e.jx[4]);
          a8[4] += w * (e.jx[1] * e.jx[2] + e.jx[4] * e.jx[5]);
          a8[5] += w * (e.jx[2] * e.jx[2] + e.jx[5] * e.jx[5]);
          a8[6] += w * (e.jx[0] * e.r0 + e.jx[3] * e.r1);
          a8[7] += w * (e.jx[1] * e.r0 + e.jx[4] * e.r1);
          a9 += w * (e.jx[2] * e.r0 + e.jx[5] * e.r1);
          if (slot != BA_NOSLOT) {
            double* sg = my + slot * BA_STAGE;
            const double sw = sqrt(w);
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
              for (int q = 0; q < 3; ++q) sg[3 * p + q] = w * (e.jc[p] * e.jx[q] + e.jc[6 + p] * e.jx[3 + q]);  // E
#pragma unroll
            for (int p = 0; p < 12; ++p) sg[36 + p] = sw * e.jc[p];
            sg[48] = sw * e.r0;
            sg[49] = sw * e.r1;
            sg[50] = (double)f;
            hdr[wave].slot_of[f] = (int8_t)slot;
          }
        }
      }
    double tot[8];
      wave_reduce8_all(a8, lane, tot);
      const double gl2 = ba_wave_sum(a9);
      gmax_acc = fmax(gmax_acc, fmax(fabs(tot[6]), fmax(fabs(tot[7]), fabs(gl2))));
      double C[6] = {tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]}, Ci[6];
      const double D0 = ba_clamp_diag(C[0]) * inv_radius, D1 = ba_clamp_diag(C[3]) * inv_radius,
                   D2 = ba_clamp_diag(C[5]) * inv_radius;
      C[0] += D0;
      C[3] += D1;
      C[5] += D2;
      ba_inv3(C, Ci);
      if (lane == 0) {  // (C + D)^-1 | g_l | D: k_ba_step's back substitution reads them instead of
        double* __restrict__ o = lmC + 12 * (size_t)lm;  // evaluating every observation once more
#pragma unroll
        for (int q = 0; q < 6; ++q) o[q] = Ci[q];
        o[6] = tot[6];
        o[7] = tot[7];
        o[8] = gl2;
        o[9] = D0;
        o[10] = D1;
        o[11] = D2;
      }
      if (Tf) {
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < Tf) {  // Y = E C^-1
          double* sg = my + lane * BA_STAGE;
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            const double e0 = sg[3 * p], e1 = sg[3 * p + 1], e2 = sg[3 * p + 2];
            sg[18 + 3 * p] = e0 * Ci[0] + e1 * Ci[1] + e2 * Ci[2];
            sg[18 + 3 * p + 1] = e0 * Ci[1] + e1 * Ci[3] + e2 * Ci[4];
            sg[18 + 3 * p + 2] = e0 * Ci[2] + e1 * Ci[4] + e2 * Ci[5];
          }
        }
        if (lane == 0) {
          hdr[wave].gl[0] = tot[6];
          hdr[wave].gl[1] = tot[7];
          hdr[wave].gl[2] = gl2;
        }
      }
    }
    if (lane == 0) hdr[wave].n_free = (int32_t)Tf;
    BA_STAMP(8);
    __syncthreads();
    BA_STAMP(9);
    // ---------------- phase B ----------------
    if (owner) {
#pragma unroll 1
      for (int q = 0; q < BA_SCHUR_WAVES; ++q) {
        if (hdr[q].n_free == 0) continue;  // uniform
        const int s = hdr[q].slot_of[bf], s2 = hdr[q].slot_of[bf2];
        if (s < 0 || s2 < 0) continue;
        const double* __restrict__ ys = stage[q] + s * BA_STAGE + 18;
        const double* __restrict__ es = stage[q] + s2 * BA_STAGE;
        double Y[18], E[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          Y[i] = ys[i];
          E[i] = es[i];
        }
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int c = 0; c < 6; ++c)
            acc[6 * p + c] -= Y[3 * p] * E[3 * c] + Y[3 * p + 1] * E[3 * c + 1] + Y[3 * p + 2] * E[3 * c + 2];
        if (diag_owner) {  // B_ff += J_c^T J_c, g_c += J_c^T r, rhs += g_c - Y g_l
          const double* __restrict__ js = stage[q] + s * BA_STAGE + 36;
          double J[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) J[i] = js[i];
          const double r0 = js[12], r1 = js[13];
          const double g0 = hdr[q].gl[0], g1 = hdr[q].gl[1], g2 = hdr[q].gl[2];
#pragma unroll
          for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[6 * p + c] += J[p] * J[c] + J[6 + p] * J[6 + c];
            const double gcp = J[p] * r0 + J[6 + p] * r1;
            vg[p] += gcp;
            vr[p] += gcp - (Y[3 * p] * g0 + Y[3 * p + 1] * g1 + Y[3 * p + 2] * g2);
            vb[p] += J[p] * J[p] + J[6 + p] * J[6 + p];
          }
        }
      }
    }
    BA_STAMP(10);
    __syncthreads();
    BA_STAMP(11);
  }
#ifdef MVO_BA_STAMPS
  if (tid == 0 && blockIdx.x == 0)
    for (int q = 8; q < 12; ++q) mvo_ba_stamps[q] += ba_acc_[q];
#endif
  // workgroup totals
  cost_acc = ba_wave_sum(cost_acc);
  gmax_acc = ba_wave_max(gmax_acc);
  if (lane == 0) {
    red[wave] = cost_acc;
    red[BA_SCHUR_WAVES + wave] = gmax_acc;
  }
  __syncthreads();
  double* __restrict__ oS = partS + (size_t)g * nS;
  double* __restrict__ oV = partV + (size_t)g * (nV + 2);
  if (owner) {
#pragma unroll
    for (int e = 0; e < 36; ++e) oS[(size_t)tid * 36 + e] = acc[e];
    if (diag_owner) {
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        oV[6 * bf + p] = vr[p];
        oV[6 * M + 6 * bf + p] = vg[p];
        oV[12 * M + 6 * bf + p] = vb[p];
      }
    }
  }
  if (tid == 0) {
    double c = 0.0, m = 0.0;
#pragma unroll
    for (int q = 0; q < BA_SCHUR_WAVES; ++q) {
      c += red[q];
      m = fmax(m, red[BA_SCHUR_WAVES + q]);
    }
    oV[nV] = c;
    oV[nV + 1] = m;
  }
}

// ---------------------------------------------------------------------------------------------
// k_ba_fixed: the landmarks WITHOUT an observation in a free frame.  They add nothing to the reduced
// camera system -- only their cost, their gradient (for the gradient-max test) and their own 3 x 3
// system (C + D)^-1, g_l, D for k_ba_step -- so they do not need k_ba_schur's one-workgroup-per-CU
// residency: lane groups of G per landmark, thousands of workgroups.  partF[g] = cost | gmax.
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ double ba_group_sum_any(double v) {
  if (G == 64) return ba_wave_sum(v);
  v += dpp_mov_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);  // row_half_mirror: the two quads of 8 lanes
  if (G >= 16) v += dpp_mov_f64<0x128>(v);  // row_ror:8
  return v;
}

template <int G, bool POSES_IN_LDS>
__global__ __launch_bounds__(BA_THREADS) void k_ba_fixed(BaParams prm, const BaState* __restrict__ st,
                                                        double radius_override, const double* __restrict__ poses2,
                                                        const double* __restrict__ X2,
                                                        const uint32_t* __restrict__ lm_off,
                                                        const uint32_t* __restrict__ obs_frame,
                                                        const double2* __restrict__ obs_uv,
                                                        const uint32_t* __restrict__ wg_pos,
                                                        const uint32_t* __restrict__ lm_list,
                                                        double* __restrict__ lmC, double* __restrict__ partF) {
  constexpr uint32_t GROUPS = BA_THREADS / G;
  extern __shared__ double lds_poses_f[];  // the current pose set (12 N doubles) when POSES_IN_LDS
  __shared__ double red[2 * BA_WAVES];
  if (st->status != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sub = tid & (G - 1), grp = tid / G;
  const int cur = st->cur;
  const double radius = radius_override > 0.0 ? radius_override : st->radius;
  const double inv_radius = 1.0 / radius;
  const double* __restrict__ poses_g = poses2 + (size_t)cur * 12 * prm.N;
  if (POSES_IN_LDS) {
    ba_stage_lds(lds_poses_f, poses_g, 12 * (uint32_t)prm.N, (uint32_t)tid, BA_THREADS);
    __syncthreads();
  }
  const double* poses = POSES_IN_LDS ? lds_poses_f : poses_g;
  const double* __restrict__ Xs = X2 + (size_t)cur * 3 * prm.L;
  const uint32_t g = blockIdx.x, pos_end = wg_pos[g + 1];
  double cost_acc = 0.0, gmax_acc = 0.0;
  for (uint32_t pos0 = wg_pos[g]; pos0 < pos_end; pos0 += GROUPS) {  // workgroup-uniform
    const uint32_t pos = pos0 + grp;
    const bool live = pos < pos_end;
    uint32_t lm = 0, o0 = 0, T = 0;
    double X[3] = {0.0, 0.0, 1.0};
    if (live) {
      lm = lm_list[pos];
      o0 = lm_off[lm];
      T = lm_off[lm + 1] - o0;
      X[0] = Xs[3 * lm];
      X[1] = Xs[3 * lm + 1];
      X[2] = Xs[3 * lm + 2];
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // C(6) gl(3)
    // BA_PREFETCH rounds of observations are in flight per lane, each in a register slot of its own
    // (consumed and refilled in place: no copies, so the compiler waits only for the slot it
    // needs).  The loop is bound by memory concurrency: 16 wavefronts per CU with one round each in
    // flight sustain ~2 TB/s.  The loads are UNCONDITIONAL, at an index clamped to the landmark's
    // last observation (the arrays hold one spare element): a load inside a divergent `if` is
    // waited for on the spot.
    const uint32_t t_last = T ? T - 1 : 0;
    uint32_t word_q[BA_PREFETCH];
    double2 uv_q[BA_PREFETCH];
#pragma unroll
    for (int d = 0; d < BA_PREFETCH; ++d) {
      const uint32_t t = sub + d * G;
      const uint32_t tc = t < t_last ? t : t_last;
      word_q[d] = obs_frame[o0 + tc];
      uv_q[d] = obs_uv[o0 + tc];
    }
    for (uint32_t t0 = sub; __ballot(t0 < T) != 0ull; t0 += BA_PREFETCH * G) {  // wavefront-uniform trip count
#pragma unroll
      for (int d = 0; d < BA_PREFETCH; ++d) {
        const uint32_t t = t0 + d * G;
        const uint32_t f = word_q[d] & 0xFFFFu;
        const double2 uv = uv_q[d];
        {
          const uint32_t tn = t + BA_PREFETCH * G;
          const uint32_t tc = tn < t_last ? tn : t_last;
          word_q[d] = obs_frame[o0 + tc];
          uv_q[d] = obs_uv[o0 + tc];
        }
        if (t < T) {
          BaObs e;
          ba_obs_eval(poses + 12 * f, X, uv.x, uv.y, prm.loss, prm.loss_scale, e);
          cost_acc += 0.5 * e.rho;
          const double w = e.w;
          acc[0] += w * (e.jx[0] * e.jx[0] + e.jx[3] * e.jx[3]);
          acc[1] += w * (e.jx[0] * e.jx[1] + e.jx[3] * e.jx[4]);
          acc[2] += w * (e.jx[0] * e.jx[2] + e.jx[3] * e.jx[5]);
          acc[3] += w * (e.jx[1] * e.jx[1] + e.jx[4] * e.jx[4]);
          acc[4] += w * (e.jx[1] * e.jx[2] + e.jx[4] * e.jx[5]);
          acc[5] += w * (e.jx[2] * e.jx[2] + e.jx[5] * e.jx[5]);
          acc[6] += w * (e.jx[0] * e.r0 + e.jx[3] * e.r1);
          acc[7] += w * (e.jx[1] * e.r0 + e.jx[4] * e.r1);
          acc[8] += w * (e.jx[2] * e.r0 + e.jx[5] * e.r1);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = ba_group_sum_any<G>(acc[q]);
    if (live) {
      double C[6] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]}, Ci[6];
      const double D0 = ba_clamp_diag(C[0]) * inv_radius, D1 = ba_clamp_diag(C[3]) * inv_radius,
                   D2 = ba_clamp_diag(C[5]) * inv_radius;
      C[0] += D0;
      C[3] += D1;
      C[5] += D2;
      ba_inv3(C, Ci);
      gmax_acc = fmax(gmax_acc, fmax(fabs(acc[6]), fmax(fabs(acc[7]), fabs(acc[8]))));
      if (sub == 0) {
        double* __restrict__ o = lmC + 12 * (size_t)lm;
#pragma unroll
        for (int q = 0; q < 6; ++q) o[q] = Ci[q];
        o[6] = acc[6];
        o[7] = acc[7];
        o[8] = acc[8];
        o[9] = D0;
        o[10] = D1;
        o[11] = D2;
      }
    }
  }
  cost_acc = ba_wave_sum(cost_acc);
  gmax_acc = ba_wave_max(gmax_acc);
  if (lane == 0) {
    red[wave] = cost_acc;
    red[BA_WAVES + wave] = gmax_acc;
  }
  __syncthreads();
  if (tid == 0) {
    partF[2 * (size_t)g] = (red[0] + red[1]) + (red[2] + red[3]);
    partF[2 * (size_t)g + 1] = fmax(fmax(red[BA_WAVES], red[BA_WAVES + 1]), fmax(red[BA_WAVES + 2], red[BA_WAVES + 3]));
  }
}

// one workgroup: k_ba_fixed's partials added up in a fixed order -> partF[2 n | 2 n + 1].  (A "last
// workgroup adds up" scheme inside k_ba_fixed needs a device-scope fence per workgroup, which on
// this chip writes the XCD's L2 back: it doubled the kernel's time.)
__global__ __launch_bounds__(BA_THREADS) void k_ba_fixed_sum(BaParams prm, const BaState* __restrict__ st,
                                                            double* __restrict__ partF) {
  __shared__ double fs[BA_THREADS], fm[BA_THREADS];
  if (st->status != 0) return;
  const uint32_t tid = threadIdx.x, n = prm.n_wg_fixed;
  double c = 0.0, m = 0.0;
  for (uint32_t q = tid; q < n; q += BA_THREADS) {
    c += partF[2 * (size_t)q];
    m = fmax(m, partF[2 * (size_t)q + 1]);
  }
  fs[tid] = c;
  fm[tid] = m;
  __syncthreads();
  for (uint32_t o = BA_THREADS / 2; o > 0; o >>= 1) {
    if (tid < o) {
      fs[tid] += fs[tid + o];
      fm[tid] = fmax(fm[tid], fm[tid + o]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    partF[2 * (size_t)n] = fs[0];
    partF[2 * (size_t)n + 1] = fm[0];
  }
}

// Sums the per-workgroup copies: a block handles 32 entries, 8 threads per entry each summing a
// contiguous slice of the workgroups (loads in flight together), slices combined in order through
// LDS -- a fixed summation order.  The last entry of V is a maximum.
__global__ __launch_bounds__(256) void k_ba_reduce(BaParams prm, const BaState* __restrict__ st,
                                                  const double* __restrict__ partS,
                                                  const double* __restrict__ partV,
                                                  const double* __restrict__ partF, double* __restrict__ redS,
                                                  double* __restrict__ redV) {
  __shared__ double sl[8][32];
  if (st->status != 0) return;
  const uint32_t nS = prm.n_blk * 36, nV = 18 * prm.M + 2;
  const uint32_t e = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const uint32_t i = blockIdx.x * 32 + e;
  const uint32_t per = (prm.n_wg + 7) / 8, g0 = slice * per, g1 = min(g0 + per, prm.n_wg);
  const bool inS = i < nS, inV = !inS && i < nS + nV;
  const bool is_max = inV && (i - nS) == nV - 1;
  const double* __restrict__ src = inS ? partS + i : partV + (i - nS);
  const size_t stride = inS ? nS : nV;
  double a = 0.0;
  if (inS || inV) {
    uint32_t g = g0;
    for (; g + 4 <= g1; g += 4) {
      const double v0 = src[(size_t)g * stride], v1 = src[(size_t)(g + 1) * stride],
                   v2 = src[(size_t)(g + 2) * stride], v3 = src[(size_t)(g + 3) * stride];
      if (is_max) a = fmax(fmax(fmax(a, v0), fmax(v1, v2)), v3);
      else a = (((a + v0) + v1) + v2) + v3;
    }
    for (; g < g1; ++g) {
      const double v = src[(size_t)g * stride];
      a = is_max ? fmax(a, v) : a + v;
    }
    if (inV && (i - nS) >= nV - 2 && prm.n_wg_fixed && slice == 0) {  // cost | gmax of k_ba_fixed's landmarks
      const double v = partF[2 * (size_t)prm.n_wg_fixed + (is_max ? 1 : 0)];
      a = is_max ? fmax(a, v) : a + v;
    }
  }
  sl[slice][e] = a;
  __syncthreads();
  if (slice == 0 && (inS || inV)) {
    double t = sl[0][e];
#pragma unroll
    for (int q = 1; q < 8; ++q) t = is_max ? fmax(t, sl[q][e]) : t + sl[q][e];
    if (inS) redS[i] = t;
    else redV[i - nS] = t;
  }
}

// R <- R Exp(w), p <- p + dp
__device__ __forceinline__ void ba_pose_plus(const double* __restrict__ in, const double d[6], double* __restrict__ out) {
  const double wx = d[0], wy = d[1], wz = d[2];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double A, B;
  if (th < 1e-8) {
    A = 1.0 - th2 / 6.0;
    B = 0.5 - th2 / 24.0;
  } else {
    A = sin(th) / th;
    B = (1.0 - cos(th)) / th2;
  }
  const double E[9] = {1.0 - B * (wy * wy + wz * wz), -A * wz + B * wx * wy, A * wy + B * wx * wz,
                       A * wz + B * wx * wy, 1.0 - B * (wx * wx + wz * wz), -A * wx + B * wy * wz,
                       -A * wy + B * wx * wz, A * wx + B * wy * wz, 1.0 - B * (wx * wx + wy * wy)};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 * i + j] = in[3 * i] * E[j] + in[3 * i + 1] * E[3 + j] + in[3 * i + 2] * E[6 + j];
#pragma unroll
  for (int k = 0; k < 3; ++k) out[9 + k] = in[9 + k] + d[3 + k];
}

// ---------------------------------------------------------------------------------------------
// k_ba_solve: one workgroup, one THREAD PER 6 x 6 BLOCK of the lower block triangle of the damped
// reduced camera system, the block held in 36 registers for the whole factorisation; the
// right-hand side rides along as block row M (forward substitution for free).  Right-looking
// block L D L^T (no square roots): per block column kb
//   (1) the diagonal thread factors its block and publishes l (unit lower), d, 1/d      -> LDS
//   (2) the panel threads (ib > kb, jb = kb) compute X = A l^-T, publish it -> LDS, keep W = X d^-1
//   (3) every trailing thread (ib >= jb > kb) subtracts W_ib X_jb^T from its registers
// -- two barriers per block column, no LDS read-modify-write.  The factor is then dumped to LDS
// and wavefront 0 runs the scalar back substitution without barriers.
// ---------------------------------------------------------------------------------------------
// value of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ double ba_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t tri(uint32_t i, uint32_t j) { return (i * (i + 1)) / 2 + j; }  // j <= i


// Back substitution and step statistics shared by both factorisation kernels.  Lf: the factor, block
// (ib, jb) at tri(ib, jb) * 36, diagonal blocks zero on and above their diagonal, block row M = the
// forward-substituted right-hand side (row 0 of each block).
__device__ __forceinline__ void ba_solve_finish(const BaParams& prm, BaState* __restrict__ st,
                                                const double* __restrict__ redV, double* __restrict__ poses2,
                                                double* __restrict__ delta_c, const double* Lf, double* xs,
                                                const double* dc2, bool bad, double gm, uint32_t M, uint32_t n,
                                                uint32_t tid) {
  if (!bad && tid < 64) {
    // Back substitution L^T x = w (w = row 0 of block row M), block row by block row from the
    // bottom, by wavefront 0 alone and without any exchange through memory.  Lane l owns components
    // 3l, 3l+1, 3l+2 in registers: block kb lives in lanes 2kb and 2kb+1, whose values every lane
    // fetches with v_readlane (wave-uniform lane numbers, compile-time register) and solves the
    // 6 x 6 unit-triangular tail itself.  The dumped diagonal blocks are zero on and above their
    // diagonal, so the update y_j -= sum_r L[6kb+r][j] x_r applied to the block's OWN components
    // leaves exactly x_j in them: one formula for every lane with jb <= kb, no selects, and LDS is
    // only read (the factor).
    const uint32_t lane = tid, jb_l = lane >> 1, jr0 = 3 * (lane & 1);
    const bool mine = 3 * lane < n;  // n = 6 M: a lane owns three components or none
    double y[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) y[q] = mine ? Lf[(size_t)tri(M, jb_l) * 36 + jr0 + q] : 0.0;
    for (int kb = (int)M - 1; kb >= 0; --kb) {
      const double* __restrict__ l = Lf + (size_t)tri(kb, kb) * 36;  // unit lower block: l[6 r + c], r > c
      // every LDS read of the step is issued up front, unconditionally (lanes without work in this
      // block row read block (kb, 0) and discard the result): one LDS latency per step
      const bool act = mine && (int)jb_l <= kb;
      const double* __restrict__ src = Lf + (size_t)tri(kb, act ? jb_l : 0u) * 36 + jr0;  // L[6 kb + r][j] = src[6 r + q]
      double lt[36], lq[3][6];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) lq[q][r] = src[6 * r + q];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int r2 = r + 1; r2 < 6; ++r2) lt[6 * r2 + r] = l[6 * r2 + r];
      double x[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) x[r] = ba_readlane(y[r % 3], 2 * kb + r / 3);
      // column-oriented: as soon as x[k] is final every row above it takes its product, and the
      // three y chains advance side by side, newest x first
      double v[3] = {y[0], y[1], y[2]};
#pragma unroll
      for (int k = 5; k >= 0; --k) {
#pragma unroll
        for (int r = 0; r < k; ++r) x[r] -= lt[6 * k + r] * x[k];
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] -= lq[q][k] * x[k];
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) y[q] = act ? v[q] : y[q];
    }
    if (mine) {
#pragma unroll
      for (int q = 0; q < 3; ++q) xs[3 * lane + q] = y[q];
    }
  }
  __syncthreads();
  if (tid < 64) {
    double mcc = 0.0, s2 = 0.0, xn2 = 0.0;
    bool nonfinite = false;
    if (!bad) {
      for (uint32_t i = tid; i < n; i += 64) {
        const double dcv = -xs[i];
        if (!isfinite(dcv)) nonfinite = true;
        delta_c[i] = dcv;
        mcc += dcv * (dc2[i] * dcv - redV[6 * M + i]);
        s2 += dcv * dcv;
      }
      const int cur = st->cur;
      if (tid < M) {
        double d[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) d[q] = -xs[6 * tid + q];
        const double* in = poses2 + ((size_t)cur * prm.N + tid) * 12;
        ba_pose_plus(in, d, poses2 + ((size_t)(1 - cur) * prm.N + tid) * 12);
        xn2 = in[9] * in[9] + in[10] * in[10] + in[11] * in[11];
      }
    }
    mcc = ba_wave_sum(mcc);
    s2 = ba_wave_sum(s2);
    xn2 = ba_wave_sum(xn2);
    const bool any_nonfinite = __any(nonfinite);
    if (tid == 0) {
      st->iteration += 1;
      st->gmax = gm;
      st->step_bad = (bad || any_nonfinite) ? 1 : 0;
      st->mcc_c = mcc;
      st->step2_c = s2;
      st->xnorm2_c = xn2;
    }
  }
}

__global__ __launch_bounds__(BA_SOLVE_THREADS) void k_ba_solve(BaParams prm, BaState* __restrict__ st,
                                                              const double* __restrict__ redS,
                                                              const double* __restrict__ redV,
                                                              double* __restrict__ poses2,
                                                              double* __restrict__ delta_c) {
  extern __shared__ double Lf[];  // (M+1)(M+2)/2 blocks x 36: the factor, block (ib, jb) at tri(ib, jb)
  // Only X is published; W_ib = X_ib d^-1 is recomputed by its readers (36 multiplies against 36 LDS
  // writes and reads).  Rows of 38 doubles = 304 B: 16-byte aligned for b128 accesses, and the rows
  // read at a per-lane block index (consecutive ib within a wavefront) start 76 banks apart, which
  // spreads 16 lanes x 4 banks over all 64; X_jb rows are mostly read by whole wavefronts at once
  __shared__ __align__(16) double pubX[2][BA_MAXF + 1][38];
  __shared__ double dl[2][36], dinvs[2][6];
  __shared__ double xs[6 * BA_MAXF], dc2[6 * BA_MAXF];
  __shared__ int bad_flag;
  if (st->status != 0) return;
  const uint32_t M = prm.M, n = 6 * M, tid = threadIdx.x;
  const double radius = st->radius;
  // gradient tolerance first: it is tested at the top of an iteration
  __shared__ double gmw[BA_SOLVE_THREADS / 64];
  {
    double v = tid < n ? fabs(redV[6 * M + tid]) : 0.0;
    v = ba_wave_max(v);
    if ((tid & 63) == 0) gmw[tid >> 6] = v;
  }
  __syncthreads();
  double gm = redV[18 * M + 1];
#pragma unroll
  for (int q = 0; q < BA_SOLVE_THREADS / 64; ++q) gm = fmax(gm, gmw[q]);
  if (gm <= prm.gtol) {
    if (tid == 0) {
      st->gmax = gm;
      st->status = 1;
      st->termination = 0;
    }
    return;
  }
#ifdef MVO_BA_STAMPS
  unsigned long long ba_acc_[16] = {0}, ba_prev_ = __builtin_amdgcn_s_memtime();
#endif
  if (tid == 0) bad_flag = 0;
  // thread -> block (ib, jb), jb <= ib <= M
  // enumerated column by column, so that the panel of a block column (the threads that work in
  // phase 2) sits in one or two wavefronts instead of being scattered over all eight
  const uint32_t nba = (M + 1) * (M + 2) / 2;
  uint32_t jb = 0;
  for (uint32_t q = 1; q <= M; ++q)
    if (tid >= q * (M + 1) - (q * (q - 1)) / 2) jb = q;
  const uint32_t ib = jb + (tid - (jb * (M + 1) - (jb * (jb - 1)) / 2));
  const bool have = tid < nba;
  double a[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) a[e] = 0.0;
  if (have) {
    if (ib == M) {  // right-hand side, row 0 of the block
#pragma unroll
      for (int c = 0; c < 6; ++c) a[c] = redV[6 * jb + c];
    } else {
      const double* __restrict__ src = redS + (size_t)ba_blk(M, jb, ib) * 36;  // stored as block (jb, ib)
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) a[6 * r + c] = src[6 * c + r];
      if (ib == jb) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const double d2 = ba_clamp_diag(redV[12 * M + 6 * ib + r]) / radius;
          a[7 * r] += d2;
          dc2[6 * ib + r] = d2;
        }
      }
    }
  }
  __syncthreads();
  BA_STAMP(0);
  for (uint32_t kb = 0; kb < M; ++kb) {
    const int buf = kb & 1;
    if (have && ib == kb && jb == kb) {  // (1) l d l^T of the diagonal block (lower part of a)
      // right-looking: column c is scaled as soon as its pivot is known and the trailing lower
      // triangle takes its rank-1 update at once -- per column the dependent chain is reciprocal,
      // scale, one FMA (the left-looking form accumulates c products before and after the reciprocal)
      bool ok = true;
      double iv[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double dcv = a[7 * c];
        if (!(dcv > 0.0) || isinf(dcv)) ok = false;
        iv[c] = ba_rcp(dcv);
        double lc[6], oc[6];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
          oc[r] = a[6 * r + c];  // l_rc d_c
          lc[r] = oc[r] * iv[c];
          a[6 * r + c] = lc[r];
        }
#pragma unroll
        for (int r = c + 1; r < 6; ++r)
#pragma unroll
          for (int q = c + 1; q <= r; ++q) a[6 * r + q] -= lc[r] * oc[q];  // l_rc d_c l_qc
      }
#pragma unroll
      for (int e = 0; e < 36; ++e) dl[buf][e] = a[e];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        dinvs[buf][c] = iv[c];
      }
      if (!ok) bad_flag = 1;
    }
    __syncthreads();
    BA_STAMP(1);
    if (bad_flag) break;  // uniform
    if (have && jb == kb && ib > kb) {  // (2) X = A l^-T (row by row), W = X d^-1
      // column-oriented (entry c final => every later entry of the row takes its product): per row a
      // dependent chain of 5 FMAs instead of 15
      double lk[36];
#pragma unroll
      for (int c = 1; c < 6; ++c)
#pragma unroll
        for (int p = 0; p < c; ++p) lk[6 * c + p] = dl[buf][6 * c + p];
#pragma unroll
      for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = p + 1; c < 6; ++c) a[6 * r + c] -= a[6 * r + p] * lk[6 * c + p];
      double2* __restrict__ px = reinterpret_cast<double2*>(pubX[buf][ib]);
#pragma unroll
      for (int e = 0; e < 18; ++e) px[e] = make_double2(a[2 * e], a[2 * e + 1]);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) a[6 * r + c] *= dinvs[buf][c];  // the factor's block: W
    }
    __syncthreads();
    BA_STAMP(2);
    if (have && jb > kb) {  // (3) A_(ib,jb) -= W_ib X_jb^T, both operands read once
      double x[36], wv[36], di[6];
      const double2* __restrict__ xj = reinterpret_cast<const double2*>(pubX[buf][jb]);
      const double2* __restrict__ xi = reinterpret_cast<const double2*>(pubX[buf][ib]);
#pragma unroll
      for (int c = 0; c < 6; ++c) di[c] = dinvs[buf][c];
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
    BA_STAMP(3);
  }
  __syncthreads();
  const bool bad = bad_flag != 0;
  if (have && !bad) {
    // diagonal blocks: strictly lower part only (see the back substitution)
#pragma unroll
    for (int e = 0; e < 36; ++e) Lf[(size_t)tri(ib, jb) * 36 + e] = (ib == jb && e % 6 >= e / 6) ? 0.0 : a[e];
  }
  __syncthreads();
  BA_STAMP(6);
  ba_solve_finish(prm, st, redV, poses2, delta_c, Lf, xs, dc2, bad, gm, M, n, tid);
  BA_STAMP(4);
#ifdef MVO_BA_STAMPS
  BA_STAMP(5);
  if (tid == 0)
    for (int q = 0; q < 8; ++q) mvo_ba_stamps[q] += ba_acc_[q];
#endif
}

// ---------------------------------------------------------------------------------------------
// k_ba_solve_mfma: the same block L D L^T with the trailing update on the matrix cores.  The lower
// triangle of the (6 M + 1) x 6 M system (last row = the right-hand side) lives in 16 x 16
// v_mfma_f64_16x16x4 accumulator tiles (<= 66 tiles, <= 9 per wavefront, 4 doubles per lane and tile;
// C layout: column = lane & 15, row = (lane >> 4) + 4 reg).  Per block column kb, two barriers:
//   (b) every thread of the three row wavefronts factors the 6 x 6 diagonal block for itself (the same
//       21 numbers, the same arithmetic: no hand-over), thread R solves ROW R of the panel against it
//       -- X = A l^-T, one row of 6 instead of a block of 6 rows -- and publishes X, -W = -X d^-1 (MFMA
//       operands, K = 6 padded to 8 with zeros) and the factor's row for the back substitution
//   (c) every tile that still has active entries takes  C -= W X^T  as two MFMAs, and
//   (a) the lanes that hold columns of the next panel, 6 kb + 6 .. 6 kb + 11, copy them to LDS
// f64 MFMA has no higher peak than the vector unit on gfx950 (64 cycles per 16x16x4 = 16 FMA per cycle
// and SIMD, scripts/exp/mfma_f64_probe.hip); what it saves is the operand traffic and issue slots of
// the thread-per-block update (2.7 k -> 1.8 k cycles per block column in isolation).  MEASURED IN PLACE
// (scripts/ba_stamps.py, M = 28): 181 k cycles per call against k_ba_solve's 174 k -- the update falls
// from 2.7 k to 2.0 k cycles per block column, but handing the panel over through LDS and the per-row
// solve (2.1 k) cost what the thread-per-block panel did (0.8 k factor + 1.55 k), and decoding the
// stored blocks into tiles costs 27 k against 11 k.  ba2: 1.83 ms per solve against 1.79.  So
// k_ba_solve stays the default; this kernel runs with MVO_HIP_BA_SOLVE_MFMA=1 (tests/test_gpu_ba.py
// holds it to the same parity bar).
// ---------------------------------------------------------------------------------------------
typedef double ba_d4 __attribute__((ext_vector_type(4)));
#define BA_NT16 ((6 * BA_MAXF + 1 + 15) / 16)
#define BA_TPW ((BA_NT16 * (BA_NT16 + 1) / 2 + BA_SOLVE_THREADS / 64 - 1) / (BA_SOLVE_THREADS / 64))

__global__ __launch_bounds__(BA_SOLVE_THREADS) void k_ba_solve_mfma(BaParams prm, BaState* __restrict__ st,
                                                                   const double* __restrict__ redS,
                                                                   const double* __restrict__ redV,
                                                                   double* __restrict__ poses2,
                                                                   double* __restrict__ delta_c) {
  extern __shared__ double Lf[];  // the factor (layout: ba_solve_finish)
  // MFMA operands, row-major, K along the row: 9 doubles per row so that the 16 rows a quarter-wave
  // reads fall into distinct banks; entries 6, 7 stay zero
  __shared__ double Wp[BA_NT16 * 16][9], Xp[BA_NT16 * 16][9];
  __shared__ double Pan[BA_NT16 * 16][6];
  __shared__ double xs[6 * BA_MAXF], dc2[6 * BA_MAXF];
  __shared__ double gmw[BA_SOLVE_THREADS / 64];
  __shared__ int bad_flag;
  if (st->status != 0) return;
  const uint32_t M = prm.M, n = 6 * M, tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const double radius = st->radius;
  {
    double v = tid < n ? fabs(redV[6 * M + tid]) : 0.0;
    v = ba_wave_max(v);
    if (lane == 0) gmw[wave] = v;
  }
  __syncthreads();
  double gm = redV[18 * M + 1];
#pragma unroll
  for (int q = 0; q < BA_SOLVE_THREADS / 64; ++q) gm = fmax(gm, gmw[q]);
  if (gm <= prm.gtol) {
    if (tid == 0) {
      st->gmax = gm;
      st->status = 1;
      st->termination = 0;
    }
    return;
  }
#ifdef MVO_BA_STAMPS
  unsigned long long ba_acc_[16] = {0}, ba_prev_ = __builtin_amdgcn_s_memtime();
#endif
  for (uint32_t i = tid; i < BA_NT16 * 16 * 9; i += BA_SOLVE_THREADS) {
    (&Wp[0][0])[i] = 0.0;
    (&Xp[0][0])[i] = 0.0;
  }
  if (tid < n) dc2[tid] = ba_clamp_diag(redV[12 * M + tid]) / radius;
  if (tid == 0) bad_flag = 0;
  // tiles straight from the system as k_ba_reduce left it (upper block triangle, 36 doubles per block,
  // L2-resident): A[R][Cc] = S[Cc][R] = entry (c, r) of block (jb, ib)
  const uint32_t NT = (n + 1 + 15) / 16, ntiles = NT * (NT + 1) / 2;
  ba_d4 acc[BA_TPW];
  int ti[BA_TPW], tj[BA_TPW];
#pragma unroll
  for (int q = 0; q < BA_TPW; ++q) {
    const uint32_t t = wave + (BA_SOLVE_THREADS / 64) * q;
    uint32_t i = 0;
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    ti[q] = t < ntiles ? (int)i : -1;
    tj[q] = (int)(t - i * (i + 1) / 2);
    const uint32_t Cc = 16 * (uint32_t)tj[q] + (lane & 15);
    const uint32_t jb = Cc / 6, c = Cc - 6 * jb;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const uint32_t R = 16 * i + (lane >> 4) + 4 * reg;
      const uint32_t ib = R / 6, r = R - 6 * ib;
      double v = 0.0;
      if (ti[q] >= 0) {
        if (R < n && Cc <= R) {
          v = redS[(size_t)ba_blk(M, jb, ib) * 36 + 6 * c + r];
          if (R == Cc) v += ba_clamp_diag(redV[12 * M + R]) / radius;
        } else if (R == n && Cc < n) {
          v = redV[Cc];
        }
      }
      acc[q][reg] = v;
    }
  }
  // panel columns c0n .. c0n + 5 (rows >= c0n) out of tile q into LDS
  auto extract = [&](int q, uint32_t c0n) {
    if (16 * (uint32_t)tj[q] > c0n + 5 || 16 * (uint32_t)tj[q] + 15 < c0n) return;  // wave-uniform
    const uint32_t cc = 16 * (uint32_t)tj[q] + (lane & 15) - c0n;  // wraps for the columns left of the panel
    if (cc < 6) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const uint32_t R = 16 * (uint32_t)ti[q] + (lane >> 4) + 4 * reg;
        if (R >= c0n && R <= n) Pan[R][cc] = acc[q][reg];
      }
    }
  };
#pragma unroll
  for (int q = 0; q < BA_TPW; ++q)
    if (ti[q] >= 0) extract(q, 0);
  __syncthreads();
  BA_STAMP(0);
  const bool row_wave = wave <= (n >> 6);  // the wavefronts that hold a row thread (tid <= n)
  for (uint32_t kb = 0; kb < M; ++kb) {
    const uint32_t c0 = 6 * kb;
    // (b) l d l^T of the diagonal block, right-looking, in every thread of the row wavefronts (the same
    // 21 numbers, the same arithmetic: no hand-over); thread R then solves row R of the panel
    if (row_wave) {
      double a[36], iv[6];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) a[6 * r + c] = Pan[c0 + r][c];
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double dcv = a[7 * c];
        if (!(dcv > 0.0) || isinf(dcv)) ok = false;
        iv[c] = ba_rcp(dcv);
        double lc[6], oc[6];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
          oc[r] = a[6 * r + c];  // l_rc d_c
          lc[r] = oc[r] * iv[c];
          a[6 * r + c] = lc[r];
        }
#pragma unroll
        for (int r = c + 1; r < 6; ++r)
#pragma unroll
          for (int q = c + 1; q <= r; ++q) a[6 * r + q] -= lc[r] * oc[q];  // l_rc d_c l_qc
      }
      if (!ok) bad_flag = 1;
      const uint32_t R = tid;
      if (R >= c0 + 6 && R <= n) {  // X = A l^-T for this row; W = X d^-1
        double x[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) x[c] = Pan[R][c];
#pragma unroll
        for (int p2 = 0; p2 < 5; ++p2)
#pragma unroll
          for (int c = p2 + 1; c < 6; ++c) x[c] -= x[p2] * a[6 * c + p2];
        double* __restrict__ lf = Lf + (size_t)tri(R / 6, kb) * 36 + 6 * (R % 6);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double w = x[c] * iv[c];
          Xp[R][c] = x[c];
          Wp[R][c] = -w;
          lf[c] = w;
        }
      } else if (R >= c0 && R < c0 + 6) {  // a row of the diagonal block: its strictly lower part is the factor; it is finished
        const uint32_t r = R - c0;
        double* __restrict__ lf = Lf + (size_t)tri(kb, kb) * 36 + 6 * r;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = 0.0;  // a[6 r + c] for c < r, by selects: r is a run-time value, a[] lives in registers
#pragma unroll
          for (int rr = c + 1; rr < 6; ++rr) v = r == (uint32_t)rr ? a[6 * rr + c] : v;
          lf[c] = v;
          Xp[R][c] = 0.0;
          Wp[R][c] = 0.0;
        }
      }
    }
    __syncthreads();
    BA_STAMP(2);
    if (bad_flag) break;  // uniform
    // (c) C -= W X^T on every tile that reaches beyond the panel; a tile that holds columns of the NEXT
    // panel hands them over while it is in registers
#pragma unroll
    for (int q = 0; q < BA_TPW; ++q) {
      if (ti[q] < 0 || 16 * (uint32_t)tj[q] + 15 < c0 + 6) continue;  // wave-uniform; ti >= tj covers the rows
      const double* __restrict__ wr = Wp[16 * ti[q] + (lane & 15)];
      const double* __restrict__ xr = Xp[16 * tj[q] + (lane & 15)];
      const double a0 = wr[lane >> 4], a1 = wr[4 + (lane >> 4)];
      const double b0 = xr[lane >> 4], b1 = xr[4 + (lane >> 4)];
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[q], 0, 0, 0);
      if (kb + 1 < M) extract(q, c0 + 6);
    }
    BA_STAMP(3);
    __syncthreads();
    BA_STAMP(1);
  }
  const bool bad = bad_flag != 0;
  __syncthreads();
  BA_STAMP(6);
  ba_solve_finish(prm, st, redV, poses2, delta_c, Lf, xs, dc2, bad, gm, M, n, tid);
  BA_STAMP(4);
#ifdef MVO_BA_STAMPS
  BA_STAMP(5);
  if (tid == 0)
    for (int q = 0; q < 8; ++q) mvo_ba_stamps[q] += ba_acc_[q];
#endif
}

// ---------------------------------------------------------------------------------------------
// k_ba_step: mode 0 = cost at the current point (start of a solve); mode 1 = landmark back
// substitution dl = -(C + D)^-1 (g_l + E^T dc), candidate landmark, cost at the candidate.
// partT[g][4] = cost | model-cost-change part | |dl|^2 | |X|^2
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BA_THREADS) void k_ba_step(BaParams prm, const BaState* __restrict__ st, int mode,
                                                       const double* __restrict__ poses2, double* __restrict__ X2,
                                                       const double* __restrict__ delta_c,
                                                       const uint32_t* __restrict__ lm_off,
                                                       const uint32_t* __restrict__ lm_nfree,
                                                       const double* __restrict__ lmC,
                                                       const uint32_t* __restrict__ obs_frame,
                                                       const double2* __restrict__ obs_uv,
                                                       const uint32_t* __restrict__ wg_lm,
                                                       double* __restrict__ partT) {
  __shared__ double red[4 * BA_WAVES];
  if (st->status != 0) return;
  if (mode == 1 && st->step_bad) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cur = st->cur;
  const double* __restrict__ poses = poses2 + (size_t)cur * 12 * prm.N;
  const double* __restrict__ posesc = poses2 + (size_t)(1 - cur) * 12 * prm.N;
  const double* __restrict__ Xs = X2 + (size_t)cur * 3 * prm.L;
  double* __restrict__ Xc = X2 + (size_t)(1 - cur) * 3 * prm.L;
  const uint32_t g = blockIdx.x;
  double cost_acc = 0.0, mcc_acc = 0.0, s2_acc = 0.0, xn2_acc = 0.0;  // the last three: wave-uniform
  for (uint32_t lm = wg_lm[g] + wave; lm < wg_lm[g + 1]; lm += BA_WAVES) {
    const uint32_t o0 = lm_off[lm], T = lm_off[lm + 1] - o0;
    const double X[3] = {Xs[3 * lm], Xs[3 * lm + 1], Xs[3 * lm + 2]};
    double Xn[3] = {X[0], X[1], X[2]};
    if (mode == 1) {
      // E^T dc over the landmark's observations in FREE frames (they come first, at most 28);
      // (C + D)^-1, g_l and D are k_ba_schur's, computed at this point with this radius
      const uint32_t Tf = lm_nfree[lm];
      double e0 = 0.0, e1 = 0.0, e2 = 0.0;
      if ((uint32_t)lane < Tf) {
        const uint32_t f = obs_frame[o0 + lane] & 0xFFFFu;
        const double2 uv = obs_uv[o0 + lane];
        BaObs e;
        ba_obs_eval(poses + 12 * f, X, uv.x, uv.y, prm.loss, prm.loss_scale, e);
        double jd0 = 0.0, jd1 = 0.0;  // E^T dc = w J_X^T (J_c dc)
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          const double dcp = delta_c[6 * f + p];
          jd0 += e.jc[p] * dcp;
          jd1 += e.jc[6 + p] * dcp;
        }
        e0 = e.w * (e.jx[0] * jd0 + e.jx[3] * jd1);
        e1 = e.w * (e.jx[1] * jd0 + e.jx[4] * jd1);
        e2 = e.w * (e.jx[2] * jd0 + e.jx[5] * jd1);
      }
      if (Tf) {
        e0 = ba_wave_sum(e0);
        e1 = ba_wave_sum(e1);
        e2 = ba_wave_sum(e2);
      }
      const double* __restrict__ c = lmC + 12 * (size_t)lm;
      const double gl[3] = {c[6], c[7], c[8]}, D[3] = {c[9], c[10], c[11]};
      const double t0 = gl[0] + e0, t1 = gl[1] + e1, t2 = gl[2] + e2;
      const double dl[3] = {-(c[0] * t0 + c[1] * t1 + c[2] * t2), -(c[1] * t0 + c[3] * t1 + c[4] * t2),
                            -(c[2] * t0 + c[4] * t1 + c[5] * t2)};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        Xn[q] = X[q] + dl[q];
        mcc_acc += dl[q] * (D[q] * dl[q] - gl[q]);
        s2_acc += dl[q] * dl[q];
        xn2_acc += X[q] * X[q];
      }
      if (lane < 3) Xc[3 * lm + lane] = lane == 0 ? Xn[0] : (lane == 1 ? Xn[1] : Xn[2]);
    }
    const double* __restrict__ pp = mode == 1 ? posesc : poses;
    for (uint32_t base = 0; base < T; base += 64) {
      const uint32_t t = base + lane;
      if (t < T) {
        const uint32_t f = obs_frame[o0 + t] & 0xFFFFu;
        const double2 uv = obs_uv[o0 + t];
        double a, b, c;
        ba_project(pp + 12 * f, Xn, a, b, c);
        const double ic = ba_rcp(c);
        const double r0 = a * ic - uv.x, r1 = b * ic - uv.y;
        double rho, w;
        ba_loss(prm.loss, prm.loss_scale, r0 * r0 + r1 * r1, rho, w);
        cost_acc += 0.5 * rho;
      }
    }
  }
  cost_acc = ba_wave_sum(cost_acc);
  if (lane == 0) {
    red[wave] = cost_acc;
    red[BA_WAVES + wave] = mcc_acc;
    red[2 * BA_WAVES + wave] = s2_acc;
    red[3 * BA_WAVES + wave] = xn2_acc;
  }
  __syncthreads();
  if (tid < 4) partT[(size_t)g * 4 + tid] = (red[4 * tid] + red[4 * tid + 1]) + (red[4 * tid + 2] + red[4 * tid + 3]);
}

// The same for SHORT tracks: a group of G = 8 or 16 lanes per landmark instead of a whole wavefront,
// 256 / G landmarks in flight per workgroup.  With some tens of observations per landmark the
// per-landmark work of the wavefront version (two 64-lane reductions, the 3 x 3 inverse) outweighs
// the per-observation work; here the reduction is 3 or 4 DPP steps shared by all groups of the
// wavefront.  Same sums per landmark; the order of the additions differs (fixed, reproducible).
template <int G>
__device__ __forceinline__ double ba_group_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);  // row_half_mirror: the two quads of 8 lanes
  if (G >= 16) v += dpp_mov_f64<0x128>(v);  // row_ror:8
  return v;
}

template <int G, bool POSES_IN_LDS>
__global__ __launch_bounds__(BA_THREADS) void k_ba_step_g(BaParams prm, const BaState* __restrict__ st, int mode,
                                                         const double* __restrict__ poses2, double* __restrict__ X2,
                                                         const double* __restrict__ delta_c,
                                                         const uint32_t* __restrict__ lm_off,
                                                         const uint32_t* __restrict__ lm_nfree,
                                                         const double* __restrict__ lmC,
                                                         const uint32_t* __restrict__ obs_frame,
                                                         const double2* __restrict__ obs_uv,
                                                         const uint32_t* __restrict__ wg_lm,
                                                         double* __restrict__ partT) {
  constexpr uint32_t GROUPS = BA_THREADS / G;
  extern __shared__ double lds_poses2[];  // both pose sets (24 N doubles) when POSES_IN_LDS
  __shared__ double red[4 * BA_WAVES];
  if (st->status != 0) return;
  if (mode == 1 && st->step_bad) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sub = tid & (G - 1), grp = tid / G;
  const int cur = st->cur;
  if (POSES_IN_LDS) {  // the pose of an observation's frame is then one LDS read behind the frame index
    ba_stage_lds(lds_poses2, poses2, 24 * (uint32_t)prm.N, (uint32_t)tid, BA_THREADS);
    __syncthreads();
  }
  const double* pbase = POSES_IN_LDS ? lds_poses2 : poses2;
  const double* poses = pbase + (size_t)cur * 12 * prm.N;
  const double* posesc = pbase + (size_t)(1 - cur) * 12 * prm.N;
  const double* __restrict__ Xs = X2 + (size_t)cur * 3 * prm.L;
  double* __restrict__ Xc = X2 + (size_t)(1 - cur) * 3 * prm.L;
  const uint32_t g = blockIdx.x, lm_end = wg_lm[g + 1];
  double cost_acc = 0.0, mcc_acc = 0.0, s2_acc = 0.0, xn2_acc = 0.0;
  for (uint32_t lm0 = wg_lm[g]; lm0 < lm_end; lm0 += GROUPS) {  // workgroup-uniform
    const uint32_t lm = lm0 + grp;
    const bool live = lm < lm_end;
    uint32_t o0 = 0, T = 0;
    double X[3] = {0.0, 0.0, 1.0};
    if (live) {
      o0 = lm_off[lm];
      T = lm_off[lm + 1] - o0;
      X[0] = Xs[3 * lm];
      X[1] = Xs[3 * lm + 1];
      X[2] = Xs[3 * lm + 2];
    }
    double Xn[3] = {X[0], X[1], X[2]};
    if (mode == 1) {
      // E^T dc over the landmark's observations in FREE frames (they come first, at most 28);
      // (C + D)^-1, g_l and D are k_ba_schur's, computed at this point with this radius
      const uint32_t Tf = live ? lm_nfree[lm] : 0u;
      double e0 = 0.0, e1 = 0.0, e2 = 0.0;
      for (uint32_t t = sub; __ballot(t < Tf) != 0ull; t += G) {  // wavefront-uniform trip count
        if (t < Tf) {
          const uint32_t f = obs_frame[o0 + t] & 0xFFFFu;
          const double2 uv = obs_uv[o0 + t];
          BaObs e;
          ba_obs_eval(poses + 12 * f, X, uv.x, uv.y, prm.loss, prm.loss_scale, e);
          double jd0 = 0.0, jd1 = 0.0;  // E^T dc = w J_X^T (J_c dc)
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            const double dcp = delta_c[6 * f + p];
            jd0 += e.jc[p] * dcp;
            jd1 += e.jc[6 + p] * dcp;
          }
          e0 += e.w * (e.jx[0] * jd0 + e.jx[3] * jd1);
          e1 += e.w * (e.jx[1] * jd0 + e.jx[4] * jd1);
          e2 += e.w * (e.jx[2] * jd0 + e.jx[5] * jd1);
        }
      }
      if (__ballot(Tf != 0) != 0ull) {
        e0 = ba_group_sum<G>(e0);
        e1 = ba_group_sum<G>(e1);
        e2 = ba_group_sum<G>(e2);
      }
      if (live) {  // every lane of the group holds the landmark's totals
        const double* __restrict__ c = lmC + 12 * (size_t)lm;
        const double gl[3] = {c[6], c[7], c[8]}, D[3] = {c[9], c[10], c[11]};
        const double t0 = gl[0] + e0, t1 = gl[1] + e1, t2 = gl[2] + e2;
        const double dl[3] = {-(c[0] * t0 + c[1] * t1 + c[2] * t2), -(c[1] * t0 + c[3] * t1 + c[4] * t2),
                              -(c[2] * t0 + c[4] * t1 + c[5] * t2)};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          Xn[q] = X[q] + dl[q];
          if (sub == 0) {  // once per landmark
            mcc_acc += dl[q] * (D[q] * dl[q] - gl[q]);
            s2_acc += dl[q] * dl[q];
            xn2_acc += X[q] * X[q];
          }
        }
        if (sub < 3) Xc[3 * lm + sub] = sub == 0 ? Xn[0] : (sub == 1 ? Xn[1] : Xn[2]);
      }
    }
    const double* pp = mode == 1 ? posesc : poses;
    {
      // BA_PREFETCH rounds of observations are in flight per lane, each in a register slot of its own
      // (consumed and refilled in place: no copies, so the compiler waits only for the slot it
      // needs).  The loop is bound by memory concurrency: 16 wavefronts per CU with one round each in
      // flight sustain ~2 TB/s.  The loads are UNCONDITIONAL, at an index clamped to the landmark's
      // last observation (the arrays hold one spare element): a load inside a divergent `if` is
      // waited for on the spot.
      const uint32_t t_last = T ? T - 1 : 0;
      uint32_t word_q[BA_PREFETCH];
      double2 uv_q[BA_PREFETCH];
#pragma unroll
      for (int d = 0; d < BA_PREFETCH; ++d) {
        const uint32_t t = sub + d * G;
        const uint32_t tc = t < t_last ? t : t_last;
        word_q[d] = obs_frame[o0 + tc];
        uv_q[d] = obs_uv[o0 + tc];
      }
      for (uint32_t t0 = sub; __ballot(t0 < T) != 0ull; t0 += BA_PREFETCH * G) {  // wavefront-uniform trip count
#pragma unroll
        for (int d = 0; d < BA_PREFETCH; ++d) {
          const uint32_t t = t0 + d * G;
          const uint32_t f = word_q[d] & 0xFFFFu;
          const double2 uv = uv_q[d];
          {
            const uint32_t tn = t + BA_PREFETCH * G;
            const uint32_t tc = tn < t_last ? tn : t_last;
            word_q[d] = obs_frame[o0 + tc];
            uv_q[d] = obs_uv[o0 + tc];
          }
          if (t < T) {
            double a, b, c;
            ba_project(pp + 12 * f, Xn, a, b, c);
            const double ic = ba_rcp(c);
            const double r0 = a * ic - uv.x, r1 = b * ic - uv.y;
            double rho, w;
            ba_loss(prm.loss, prm.loss_scale, r0 * r0 + r1 * r1, rho, w);
            cost_acc += 0.5 * rho;
          }
        }
      }
    }
  }
  cost_acc = ba_wave_sum(cost_acc);
  mcc_acc = ba_wave_sum(mcc_acc);
  s2_acc = ba_wave_sum(s2_acc);
  xn2_acc = ba_wave_sum(xn2_acc);
  if (lane == 0) {
    red[wave] = cost_acc;
    red[BA_WAVES + wave] = mcc_acc;
    red[2 * BA_WAVES + wave] = s2_acc;
    red[3 * BA_WAVES + wave] = xn2_acc;
  }
  __syncthreads();
  if (tid < 4) partT[(size_t)g * 4 + tid] = (red[4 * tid] + red[4 * tid + 1]) + (red[4 * tid + 2] + red[4 * tid + 3]);
}

// partitioned mode: this rank's four sums in one place, ready for the caller's all-reduce
__global__ __launch_bounds__(64) void k_ba_sum4(BaParams prm, const BaState* __restrict__ st, int mode,
                                               const double* __restrict__ partT, double* __restrict__ sums4) {
  if (st->status != 0) return;
  const int lane = threadIdx.x;
  double v[4] = {0, 0, 0, 0};
  if (mode == 0 || !st->step_bad)
    for (uint32_t g = lane; g < prm.n_wg_step; g += 64)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += partT[(size_t)g * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = ba_wave_sum(v[k]);
  if (lane < 4) sums4[lane] = lane == 0 ? v[0] : (lane == 1 ? v[1] : (lane == 2 ? v[2] : v[3]));
}

// ---------------------------------------------------------------------------------------------
// k_ba_decide: the trust-region rules of include/mvo_hip_ba.h on the device-resident state
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ba_decide(BaParams prm, BaState* __restrict__ st, int mode,
                                                 const double* __restrict__ partT, BaTrace* __restrict__ trace) {
  if (st->status != 0) return;
  const int lane = threadIdx.x;
  double v[4] = {0, 0, 0, 0};
  const bool have = mode == 0 || !st->step_bad;
  if (have)
    for (uint32_t g = lane; g < prm.n_wg_step; g += 64)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += partT[(size_t)g * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = ba_wave_sum(v[k]);
  if (lane != 0) return;
  BaState s = *st;
  BaTrace it;
  it.cost = it.cost_change = it.gradient_max_norm = it.step_norm = it.relative_decrease = 0.0;
  it.trust_region_radius = 0.0;
  it.step_is_valid = it.step_is_successful = 0;
  if (mode == 0) {
    s.cost = s.initial_cost = v[0];
    it.cost = s.cost;
    it.trust_region_radius = s.radius;
    it.step_is_valid = it.step_is_successful = 1;
    if (trace && s.ntrace < (int)prm.trace_cap) trace[s.ntrace] = it;
    ++s.ntrace;
    if (prm.max_it <= 0) {
      s.status = 1;
      s.termination = 1;
    }
    *st = s;
    return;
  }
  it.gradient_max_norm = s.gmax;
  const double mcc = 0.5 * (s.mcc_c + v[1]);
  const double step2 = s.step2_c + v[2];
  bool bad = s.step_bad || !isfinite(step2) || !(mcc > 0.0);
  if (bad) {
    if (++s.num_invalid >= 5) {
      s.status = 1;
      s.termination = 2;
    } else {
      s.radius *= 0.5;
      it.cost = s.cost;
      it.trust_region_radius = s.radius;
      ++s.nunsucc;
      if (trace && s.ntrace < (int)prm.trace_cap) trace[s.ntrace] = it;
      ++s.ntrace;
      if (s.iteration >= prm.max_it) {
        s.status = 1;
        s.termination = 1;
      }
    }
    *st = s;
    return;
  }
  s.num_invalid = 0;
  it.step_is_valid = 1;
  it.step_norm = sqrt(step2);
  if (it.step_norm <= prm.ptol * (sqrt(s.xnorm2_c + v[3]) + prm.ptol)) {
    s.status = 1;
    s.termination = 0;
    s.iteration -= 1;
    *st = s;
    return;
  }
  const double cost_c = v[0];
  it.cost_change = s.cost - cost_c;
  it.relative_decrease = it.cost_change / mcc;
  const double cost_before = s.cost;
  if (it.relative_decrease > 1e-3) {
    s.cur ^= 1;
    s.cost = cost_c;
    const double q = 2.0 * it.relative_decrease - 1.0;
    s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
    s.radius = fmin(1e16, s.radius);
    s.decf = 2.0;
    it.step_is_successful = 1;
    ++s.nsucc;
  } else {
    s.radius = s.radius / s.decf;
    s.decf *= 2.0;
    ++s.nunsucc;
  }
  it.cost = s.cost;
  it.trust_region_radius = s.radius;
  if (trace && s.ntrace < (int)prm.trace_cap) trace[s.ntrace] = it;
  ++s.ntrace;
  if (it.step_is_successful && fabs(it.cost_change) <= prm.ftol * cost_before) {
    s.status = 1;
    s.termination = 0;
  } else if (s.radius <= 1e-32) {
    s.status = 1;
    s.termination = 0;
  } else if (s.iteration >= prm.max_it) {
    s.status = 1;
    s.termination = 1;
  }
  *st = s;
}

// debug: residuals of every observation at the current point, caller's observation order
__global__ __launch_bounds__(256) void k_ba_residuals(BaParams prm, uint64_t n_obs, const double* __restrict__ poses,
                                                     const double* __restrict__ Xs,
                                                     const uint32_t* __restrict__ obs_lm,
                                                     const uint32_t* __restrict__ obs_frame,
                                                     const double2* __restrict__ obs_uv,
                                                     const uint32_t* __restrict__ obs_src,
                                                     double* __restrict__ out) {
  const uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_obs) return;
  const uint32_t lm = obs_lm[o], f = obs_frame[o] & 0xFFFFu;
  const double X[3] = {Xs[3 * lm], Xs[3 * lm + 1], Xs[3 * lm + 2]};
  double a, b, c;
  ba_project(poses + 12 * f, X, a, b, c);
  const double2 uv = obs_uv[o];
  const uint32_t dst = obs_src[o];
  out[2 * (size_t)dst] = a / c - uv.x;
  out[2 * (size_t)dst + 1] = b / c - uv.y;
}

// =============================================================================================
// host side
// =============================================================================================
namespace {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)mvo_device_free(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = mvo_device_alloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)mvo_device_free(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

constexpr int kMaxSchurEvents = 64;

}  // namespace

struct mvo_hip_ba_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipError_t last_hip_error = hipSuccess;
  Buf d_poses, d_X, d_lm_off, d_lm_nfree, d_obs_frame, d_obs_uv, d_obs_lm, d_obs_src, d_wg_lm, d_wg_lm_step, d_partS, d_partV,
      d_redS, d_redV, d_partT, d_delta, d_state, d_trace, d_res, d_sums4, d_lmC, d_list_free, d_list_fixed, d_wg_fixed,
      d_partF;
  bool split = false;  // landmarks without free observations go through k_ba_fixed
  bool uploaded = false;
  BaParams prm{};
  uint64_t n_obs = 0;
  int step_group = 64;  // lanes per landmark in k_ba_step: 8 / 16 for short tracks, a wavefront otherwise
  std::vector<mvo_hip_ba_pose> h_poses0;  // the uploaded point (solve restarts from it)
  std::vector<double> h_X0;
  std::vector<mvo_hip_ba_pose> h_poses;   // result of the last solve
  std::vector<double> h_X;
  BaState h_state{};
  bool solved = false;
  int32_t trace_cap = 0;
  bool profiling = false;
  hipEvent_t ev[2 + 2 * kMaxSchurEvents] = {};
  bool events_ready = false;
  mvo_hip_ba_timing timing{};
  size_t solve_lds = 0;
  bool solve_mfma = false;  // MVO_HIP_BA_SOLVE_MFMA=1 at upload: k_ba_solve_mfma (measured 3 % slower, kept for the record)
  uint64_t schur_flops = 0;
};

namespace {

#define BA_TRY(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      h->last_hip_error = e_;                                                             \
      return e_ == hipErrorOutOfMemory ? MVO_HIP_ERR_OUT_OF_MEMORY : MVO_HIP_ERR_DEVICE; \
    }                                                                                     \
  } while (0)

int ba_upload(mvo_hip_ba_context* h, const mvo_hip_ba_pose* poses, uint32_t M, uint32_t N, const double* X,
              const mvo_hip_ba_observations* obs, const mvo_hip_ba_options* opt) {
  const auto t0 = std::chrono::steady_clock::now();
  if (!poses || !X || !obs || !opt || !obs->landmark_offsets) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (M == 0 || N <= M) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (M > MVO_HIP_BA_MAX_FREE_FRAMES || N > MVO_HIP_BA_MAX_FRAMES) return MVO_HIP_ERR_UNSUPPORTED;
  if (opt->max_num_iterations < 0 || opt->loss < 0 || opt->loss > 2) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (opt->loss != 0 && !(opt->loss_scale > 0.0)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  const uint32_t L = obs->n_landmarks;
  if (L == 0 || obs->landmark_offsets[0] != 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  for (uint32_t j = 0; j < L; ++j)
    if (obs->landmark_offsets[j + 1] < obs->landmark_offsets[j]) return MVO_HIP_ERR_INVALID_ARGUMENT;
  const uint64_t n_obs = obs->landmark_offsets[L];
  if (n_obs >= (1ull << 31)) return MVO_HIP_ERR_UNSUPPORTED;
  if (n_obs && (!obs->frame || !obs->rays)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  BA_TRY(hipSetDevice(h->device));
  h->uploaded = false;
  h->solved = false;

  // per-landmark observation lists sorted by frame; free observations get consecutive slots
  std::vector<uint32_t> lm_off(L + 1), lm_nfree(L), frame_word(n_obs), obs_lm(n_obs), obs_src(n_obs);
  std::vector<double> uv(2 * n_obs);
  std::vector<uint32_t> order;
  for (uint32_t j = 0; j < L; ++j) {
    const uint64_t o0 = obs->landmark_offsets[j], o1 = obs->landmark_offsets[j + 1];
    lm_off[j] = (uint32_t)o0;
    order.resize(o1 - o0);
    for (uint64_t o = o0; o < o1; ++o) {
      if (obs->frame[o] >= N) return MVO_HIP_ERR_INVALID_ARGUMENT;
      if (!(obs->rays[3 * o + 2] > 0.0)) return MVO_HIP_ERR_INVALID_ARGUMENT;
      order[o - o0] = (uint32_t)o;
    }
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return obs->frame[a] < obs->frame[b]; });
    uint32_t slot = 0;
    for (uint64_t k = 0; k < o1 - o0; ++k) {
      const uint32_t src = order[k], f = obs->frame[src];
      const uint64_t dst = o0 + k;
      uint32_t sl = BA_NOSLOT;
      if (f < M && k > 0 && obs->frame[order[k - 1]] == f) return MVO_HIP_ERR_UNSUPPORTED;  // one observation per (landmark, frame)
      if (f < M) {
        if (slot >= BA_MAXF) return MVO_HIP_ERR_UNSUPPORTED;  // > 28 observations of one landmark in free frames
        sl = slot++;
      }
      frame_word[dst] = f | (sl << 16);
      obs_lm[dst] = j;
      obs_src[dst] = src;
      uv[2 * dst] = obs->rays[3 * (size_t)src] / obs->rays[3 * (size_t)src + 2];
      uv[2 * dst + 1] = obs->rays[3 * (size_t)src + 1] / obs->rays[3 * (size_t)src + 2];
    }
    lm_nfree[j] = slot;
  }
  lm_off[L] = (uint32_t)n_obs;
  {
    uint64_t fl = 96ull * n_obs;
    for (uint32_t j = 0; j < L; ++j) fl += 380ull * lm_nfree[j] + 216ull * ((uint64_t)lm_nfree[j] * (lm_nfree[j] + 1) / 2);
    h->schur_flops = fl;
  }

  // landmarks -> workgroups: contiguous ranges of about equal load (observations + one unit per
  // landmark).  k_ba_schur holds the reduced camera system per workgroup: one workgroup per CU;
  // k_ba_step has no such state and is cut finer, for occupancy.
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
  auto partition = [&](uint32_t n_wg) {
    std::vector<uint32_t> w(n_wg + 1, L);
    w[0] = 0;
    uint32_t g = 1;
    for (uint32_t j = 0; j < L && g < n_wg; ++j)
      while (g < n_wg && (uint64_t)lm_off[j + 1] + (j + 1) >= (n_obs + L) * g / n_wg) w[g++] = j + 1;
    for (uint32_t q = 1; q <= n_wg; ++q)
      if (w[q] < w[q - 1]) w[q] = w[q - 1];
    return w;
  };
  const uint32_t n_wg = (uint32_t)std::min<uint64_t>((uint64_t)n_cu, std::max<uint64_t>(1, (L + 2 * BA_SCHUR_WAVES - 1) / (2 * BA_SCHUR_WAVES)));  // without the split below
  // lanes per landmark in k_ba_step by the mean track length; workgroups of at least two rounds of landmarks
  const uint64_t mean_T = L ? n_obs / L : 0;
  h->step_group = mean_T <= 96 ? 8 : (mean_T <= 224 ? 16 : 64);
  if (const char* e = std::getenv("MVO_BA_STEP_GROUP")) h->step_group = std::atoi(e);  // tuning knob: 8 | 16 | 64
  const uint32_t per_round = h->step_group == 64 ? BA_WAVES : BA_THREADS / h->step_group;
  const uint32_t n_wg_step = (uint32_t)std::min<uint64_t>(8ull * n_cu, std::max<uint64_t>(1, (L + 2 * per_round - 1) / (2 * per_round)));
  const std::vector<uint32_t> wg_lm_step = partition(n_wg_step);
  // Landmarks without an observation in a free frame leave the reduced camera system alone: when
  // they are a sizeable share they go through k_ba_fixed (many small workgroups) and k_ba_schur
  // walks the list of the others.
  std::vector<uint32_t> list_free, list_fixed;
  for (uint32_t j = 0; j < L; ++j) (lm_nfree[j] ? list_free : list_fixed).push_back(j);
  const bool split = list_fixed.size() * 10 >= (size_t)L && !list_fixed.empty();
  h->split = split;
  auto partition_list = [&](const std::vector<uint32_t>& list, uint32_t parts) {  // positions, equal observations + 1 each
    std::vector<uint32_t> w(parts + 1, (uint32_t)list.size());
    w[0] = 0;
    uint64_t total = 0;
    for (uint32_t j : list) total += (lm_off[j + 1] - lm_off[j]) + 1;
    uint64_t run = 0;
    uint32_t gq = 1;
    for (uint32_t k = 0; k < list.size() && gq < parts; ++k) {
      run += (lm_off[list[k] + 1] - lm_off[list[k]]) + 1;
      while (gq < parts && run >= total * gq / parts) w[gq++] = k + 1;
    }
    for (uint32_t q = 1; q <= parts; ++q)
      if (w[q] < w[q - 1]) w[q] = w[q - 1];
    return w;
  };
  uint32_t n_wg_fixed = 0;
  std::vector<uint32_t> wg_lm, wg_fixed;
  uint32_t n_wg_eff = n_wg;
  if (split) {
    n_wg_eff = (uint32_t)std::min<uint64_t>((uint64_t)n_cu, std::max<uint64_t>(1, (list_free.size() + 2 * BA_SCHUR_WAVES - 1) / (2 * BA_SCHUR_WAVES)));
    wg_lm = partition_list(list_free, n_wg_eff);
    const uint32_t per_round_f = h->step_group == 64 ? BA_WAVES : BA_THREADS / h->step_group;
    n_wg_fixed = (uint32_t)std::min<uint64_t>(8ull * n_cu, std::max<uint64_t>(1, (list_fixed.size() + 2 * per_round_f - 1) / (2 * per_round_f)));
    wg_fixed = partition_list(list_fixed, n_wg_fixed);
  } else {
    wg_lm = partition(n_wg);
  }

  BaParams& prm = h->prm;
  prm.ftol = opt->function_tolerance;
  prm.gtol = opt->gradient_tolerance;
  prm.ptol = opt->parameter_tolerance;
  prm.loss_scale = opt->loss_scale;
  prm.initial_radius = opt->initial_radius > 0.0 ? opt->initial_radius : 1e4;
  prm.max_it = opt->max_num_iterations;
  prm.loss = opt->loss;
  prm.M = (int32_t)M;
  prm.N = (int32_t)N;
  prm.L = L;
  prm.n_blk = M * (M + 1) / 2;
  prm.n_wg = n_wg_eff;
  prm.n_wg_fixed = n_wg_fixed;
  prm.n_wg_step = n_wg_step;
  prm.trace_cap = (uint32_t)h->trace_cap;
  h->n_obs = n_obs;
  h->h_poses0.assign(poses, poses + N);
  h->h_X0.assign(X, X + 3 * (size_t)L);

  const size_t nS = (size_t)prm.n_blk * 36, nV = 18 * (size_t)M + 2;
  BA_TRY(h->d_poses.reserve(sizeof(double) * 24 * N));
  BA_TRY(h->d_X.reserve(sizeof(double) * 6 * (size_t)L));
  BA_TRY(h->d_lm_off.reserve(sizeof(uint32_t) * (L + 1)));
  BA_TRY(h->d_lm_nfree.reserve(sizeof(uint32_t) * L));
  BA_TRY(h->d_obs_frame.reserve(sizeof(uint32_t) * (n_obs + 1)));
  BA_TRY(h->d_obs_uv.reserve(sizeof(double) * 2 * (n_obs + 1)));
  BA_TRY(h->d_obs_lm.reserve(sizeof(uint32_t) * (n_obs + 1)));
  BA_TRY(h->d_obs_src.reserve(sizeof(uint32_t) * (n_obs + 1)));
  BA_TRY(h->d_wg_lm.reserve(sizeof(uint32_t) * (n_wg_eff + 1)));
  BA_TRY(h->d_list_free.reserve(sizeof(uint32_t) * (list_free.size() + 1)));
  BA_TRY(h->d_list_fixed.reserve(sizeof(uint32_t) * (list_fixed.size() + 1)));
  BA_TRY(h->d_wg_fixed.reserve(sizeof(uint32_t) * (n_wg_fixed + 2)));
  BA_TRY(h->d_partF.reserve(sizeof(double) * 2 * (n_wg_fixed + 1)));
  BA_TRY(h->d_wg_lm_step.reserve(sizeof(uint32_t) * (n_wg_step + 1)));
  BA_TRY(h->d_partS.reserve(sizeof(double) * nS * n_wg_eff));
  BA_TRY(h->d_partV.reserve(sizeof(double) * nV * n_wg_eff));
  BA_TRY(h->d_redS.reserve(sizeof(double) * nS));
  BA_TRY(h->d_redV.reserve(sizeof(double) * nV));
  BA_TRY(h->d_partT.reserve(sizeof(double) * 4 * n_wg_step));
  BA_TRY(h->d_lmC.reserve(sizeof(double) * 12 * (size_t)L));
  BA_TRY(h->d_delta.reserve(sizeof(double) * 6 * M));
  BA_TRY(h->d_state.reserve(sizeof(BaState)));
  hipStream_t s = h->stream;
  BA_TRY(hipMemcpyAsync(h->d_lm_off.p, lm_off.data(), sizeof(uint32_t) * (L + 1), hipMemcpyHostToDevice, s));
  BA_TRY(hipMemcpyAsync(h->d_lm_nfree.p, lm_nfree.data(), sizeof(uint32_t) * L, hipMemcpyHostToDevice, s));
  BA_TRY(hipMemcpyAsync(h->d_wg_lm.p, wg_lm.data(), sizeof(uint32_t) * (n_wg_eff + 1), hipMemcpyHostToDevice, s));
  if (split) {
    BA_TRY(hipMemcpyAsync(h->d_list_free.p, list_free.data(), sizeof(uint32_t) * list_free.size(), hipMemcpyHostToDevice, s));
    BA_TRY(hipMemcpyAsync(h->d_list_fixed.p, list_fixed.data(), sizeof(uint32_t) * list_fixed.size(), hipMemcpyHostToDevice, s));
    BA_TRY(hipMemcpyAsync(h->d_wg_fixed.p, wg_fixed.data(), sizeof(uint32_t) * (n_wg_fixed + 1), hipMemcpyHostToDevice, s));
  }
  BA_TRY(hipMemcpyAsync(h->d_wg_lm_step.p, wg_lm_step.data(), sizeof(uint32_t) * (n_wg_step + 1), hipMemcpyHostToDevice, s));
  if (n_obs) {
    BA_TRY(hipMemcpyAsync(h->d_obs_frame.p, frame_word.data(), sizeof(uint32_t) * n_obs, hipMemcpyHostToDevice, s));
    BA_TRY(hipMemcpyAsync(h->d_obs_uv.p, uv.data(), sizeof(double) * 2 * n_obs, hipMemcpyHostToDevice, s));
    BA_TRY(hipMemcpyAsync(h->d_obs_lm.p, obs_lm.data(), sizeof(uint32_t) * n_obs, hipMemcpyHostToDevice, s));
    BA_TRY(hipMemcpyAsync(h->d_obs_src.p, obs_src.data(), sizeof(uint32_t) * n_obs, hipMemcpyHostToDevice, s));
  }
  BA_TRY(hipStreamSynchronize(s));

  BA_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ba_schur<true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 512 * (int)sizeof(double)));
  h->solve_lds = sizeof(double) * 36 * ((size_t)(M + 1) * (M + 2) / 2);
  BA_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ba_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)h->solve_lds));
  BA_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ba_solve_mfma),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds));
  h->solve_mfma = std::getenv("MVO_HIP_BA_SOLVE_MFMA") != nullptr;
  h->uploaded = true;
  h->timing.upload_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return MVO_HIP_OK;
}

// (re)load the uploaded point into both parameter buffers and reset the solver state
int ba_reset_point(mvo_hip_ba_context* h) {
  hipStream_t s = h->stream;
  const uint32_t N = h->prm.N, L = h->prm.L;
  for (int c = 0; c < 2; ++c) {
    BA_TRY(hipMemcpyAsync(h->d_poses.as<double>() + (size_t)c * 12 * N, h->h_poses0.data(), sizeof(double) * 12 * N,
                          hipMemcpyHostToDevice, s));
    BA_TRY(hipMemcpyAsync(h->d_X.as<double>() + (size_t)c * 3 * L, h->h_X0.data(), sizeof(double) * 3 * L,
                          hipMemcpyHostToDevice, s));
  }
  BaState st{};
  st.radius = h->prm.initial_radius;
  st.decf = 2.0;
  h->h_state = st;
  BA_TRY(hipMemcpyAsync(h->d_state.p, &h->h_state, sizeof(BaState), hipMemcpyHostToDevice, s));
  BA_TRY(hipStreamSynchronize(s));  // h_state is reused as the read-back target
  return MVO_HIP_OK;
}

// the camera solve: one workgroup
hipError_t launch_solve(mvo_hip_ba_context* h, hipStream_t s) {
  if (h->solve_mfma)
    hipLaunchKernelGGL(k_ba_solve_mfma, dim3(1), dim3(BA_SOLVE_THREADS), h->solve_lds, s, h->prm,
                       h->d_state.as<BaState>(), h->d_redS.as<double>(), h->d_redV.as<double>(),
                       h->d_poses.as<double>(), h->d_delta.as<double>());
  else
    hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(BA_SOLVE_THREADS), h->solve_lds, s, h->prm,
                       h->d_state.as<BaState>(), h->d_redS.as<double>(), h->d_redV.as<double>(),
                       h->d_poses.as<double>(), h->d_delta.as<double>());
  return hipGetLastError();
}

hipError_t launch_schur(mvo_hip_ba_context* h, double radius_override) {
  // poses of up to 512 frames are staged in LDS (48 KB); larger windows read them through L2
  const bool in_lds = h->prm.N <= 512;
  const size_t dyn = in_lds ? sizeof(double) * 12 * (size_t)h->prm.N : 0;
  if (h->split) {
    auto fk = &k_ba_fixed<64, false>;
    if (h->step_group == 8) fk = in_lds ? &k_ba_fixed<8, true> : &k_ba_fixed<8, false>;
    else if (h->step_group == 16) fk = in_lds ? &k_ba_fixed<16, true> : &k_ba_fixed<16, false>;
    else if (in_lds) fk = &k_ba_fixed<64, true>;
    hipLaunchKernelGGL(fk, dim3(h->prm.n_wg_fixed), dim3(BA_THREADS), dyn, h->stream, h->prm, h->d_state.as<BaState>(),
                       radius_override, h->d_poses.as<double>(), h->d_X.as<double>(), h->d_lm_off.as<uint32_t>(),
                       h->d_obs_frame.as<uint32_t>(), h->d_obs_uv.as<double2>(), h->d_wg_fixed.as<uint32_t>(),
                       h->d_list_fixed.as<uint32_t>(), h->d_lmC.as<double>(), h->d_partF.as<double>());
    hipLaunchKernelGGL(k_ba_fixed_sum, dim3(1), dim3(BA_THREADS), 0, h->stream, h->prm, h->d_state.as<BaState>(),
                       h->d_partF.as<double>());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  auto kern = in_lds ? &k_ba_schur<true> : &k_ba_schur<false>;
  hipLaunchKernelGGL(kern, dim3(h->prm.n_wg), dim3(BA_SCHUR_THREADS), dyn, h->stream, h->prm,
                     h->d_state.as<BaState>(), radius_override, h->d_poses.as<double>(), h->d_X.as<double>(),
                     h->d_lm_off.as<uint32_t>(), h->d_lm_nfree.as<uint32_t>(), h->d_obs_frame.as<uint32_t>(),
                     h->d_obs_uv.as<double2>(), h->d_wg_lm.as<uint32_t>(),
                     h->split ? h->d_list_free.as<uint32_t>() : (const uint32_t*)nullptr, h->d_partS.as<double>(),
                     h->d_partV.as<double>(), h->d_lmC.as<double>());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const uint32_t n = h->prm.n_blk * 36 + 18 * h->prm.M + 2;
  hipLaunchKernelGGL(k_ba_reduce, dim3((n + 31) / 32), dim3(256), 0, h->stream, h->prm, h->d_state.as<BaState>(),
                     h->d_partS.as<double>(), h->d_partV.as<double>(), h->d_partF.as<double>(), h->d_redS.as<double>(),
                     h->d_redV.as<double>());
  return hipGetLastError();
}

hipError_t launch_step_only(mvo_hip_ba_context* h, int mode) {
  const size_t pose_bytes = sizeof(double) * 24 * h->prm.N;
  const bool in_lds = h->step_group != 64 && pose_bytes <= 48 * 1024;
  auto kern = &k_ba_step;
  if (h->step_group == 8) kern = in_lds ? &k_ba_step_g<8, true> : &k_ba_step_g<8, false>;
  else if (h->step_group == 16) kern = in_lds ? &k_ba_step_g<16, true> : &k_ba_step_g<16, false>;
  hipLaunchKernelGGL(kern, dim3(h->prm.n_wg_step), dim3(BA_THREADS), in_lds ? pose_bytes : 0, h->stream, h->prm,
                     h->d_state.as<BaState>(), mode, h->d_poses.as<double>(), h->d_X.as<double>(),
                     h->d_delta.as<double>(), h->d_lm_off.as<uint32_t>(), h->d_lm_nfree.as<uint32_t>(),
                     h->d_lmC.as<double>(), h->d_obs_frame.as<uint32_t>(), h->d_obs_uv.as<double2>(),
                     h->d_wg_lm_step.as<uint32_t>(), h->d_partT.as<double>());
  return hipGetLastError();
}

hipError_t launch_step(mvo_hip_ba_context* h, int mode) {
  hipError_t e = launch_step_only(h, mode);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_ba_decide, dim3(1), dim3(64), 0, h->stream, h->prm, h->d_state.as<BaState>(), mode,
                     h->d_partT.as<double>(), h->d_trace.as<BaTrace>());
  return hipGetLastError();
}

// ---- landmark-partitioned mode: every rank holds all poses and a range of the landmarks; the
// reduced camera system and the step sums are additive over landmarks, so the caller all-reduces
// them between these calls and every rank takes the identical LM step.  Each call drains the stream.
int ba_fetch_results(mvo_hip_ba_context* h) {
  hipStream_t s = h->stream;
  const uint32_t N = h->prm.N, L = h->prm.L;
  const int cur = h->h_state.cur;
  h->h_poses.resize(N);
  h->h_X.resize(3 * (size_t)L);
  BA_TRY(hipMemcpyAsync(h->h_poses.data(), h->d_poses.as<double>() + (size_t)cur * 12 * N, sizeof(double) * 12 * N,
                        hipMemcpyDeviceToHost, s));
  BA_TRY(hipMemcpyAsync(h->h_X.data(), h->d_X.as<double>() + (size_t)cur * 3 * L, sizeof(double) * 3 * L,
                        hipMemcpyDeviceToHost, s));
  BA_TRY(hipStreamSynchronize(s));
  h->solved = true;
  return MVO_HIP_OK;
}

int ba_partial_sum4(mvo_hip_ba_context* h, int mode) {
  BA_TRY(h->d_sums4.reserve(sizeof(double) * 4));
  hipLaunchKernelGGL(k_ba_sum4, dim3(1), dim3(64), 0, h->stream, h->prm, h->d_state.as<BaState>(), mode,
                     h->d_partT.as<double>(), h->d_sums4.as<double>());
  BA_TRY(hipGetLastError());
  BA_TRY(hipStreamSynchronize(h->stream));
  return MVO_HIP_OK;
}

int ba_partial_begin(mvo_hip_ba_context* h) {
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  BA_TRY(hipSetDevice(h->device));
  h->prm.trace_cap = (uint32_t)h->trace_cap;
  if (h->trace_cap > 0) BA_TRY(h->d_trace.reserve(sizeof(BaTrace) * (size_t)h->trace_cap));
  int rc = ba_reset_point(h);
  if (rc != MVO_HIP_OK) return rc;
  h->solved = false;
  BA_TRY(launch_step_only(h, 0));
  return ba_partial_sum4(h, 0);
}

int ba_partial_linearize(mvo_hip_ba_context* h) {
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  BA_TRY(hipSetDevice(h->device));
  BA_TRY(launch_schur(h, 0.0));
  BA_TRY(hipStreamSynchronize(h->stream));
  return MVO_HIP_OK;
}

int ba_partial_solve(mvo_hip_ba_context* h) {
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  BA_TRY(hipSetDevice(h->device));
  BA_TRY(launch_solve(h, h->stream));
  BA_TRY(launch_step_only(h, 1));
  return ba_partial_sum4(h, 1);
}

int ba_partial_decide(mvo_hip_ba_context* h, int mode, int32_t* done) {
  if (!h->uploaded || !h->d_sums4.p) return MVO_HIP_ERR_STATE;
  BA_TRY(hipSetDevice(h->device));
  BaParams one = h->prm;
  one.n_wg_step = 1;  // the four sums are already complete (this rank's, or all ranks' after the all-reduce)
  hipLaunchKernelGGL(k_ba_decide, dim3(1), dim3(64), 0, h->stream, one, h->d_state.as<BaState>(), mode,
                     h->d_sums4.as<double>(), h->d_trace.as<BaTrace>());
  BA_TRY(hipGetLastError());
  BA_TRY(hipMemcpyAsync(&h->h_state, h->d_state.p, sizeof(BaState), hipMemcpyDeviceToHost, h->stream));
  BA_TRY(hipStreamSynchronize(h->stream));
  const bool fin = h->h_state.status != 0;
  if (done) *done = fin ? 1 : 0;
  return fin ? ba_fetch_results(h) : MVO_HIP_OK;
}

int ba_solve(mvo_hip_ba_context* h) {
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  const auto t0 = std::chrono::steady_clock::now();
  BA_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  h->prm.trace_cap = (uint32_t)h->trace_cap;
  if (h->trace_cap > 0) BA_TRY(h->d_trace.reserve(sizeof(BaTrace) * (size_t)h->trace_cap));
  const bool prof = h->profiling;
  if (prof && !h->events_ready) {
    for (auto& e : h->ev) BA_TRY(hipEventCreate(&e));
    h->events_ready = true;
  }
  int rc = ba_reset_point(h);
  if (rc != MVO_HIP_OK) return rc;
  int n_schur = 0;
  if (prof) BA_TRY(hipEventRecord(h->ev[0], s));
  BA_TRY(launch_step(h, 0));
  const int chunk = 4;
  for (int issued = 0; issued < h->prm.max_it;) {
    const int n = std::min(chunk, h->prm.max_it - issued);
    for (int i = 0; i < n; ++i) {
      const bool timed = prof && n_schur < kMaxSchurEvents;
      if (timed) BA_TRY(hipEventRecord(h->ev[2 + 2 * n_schur], s));
      BA_TRY(launch_schur(h, 0.0));
      if (timed) BA_TRY(hipEventRecord(h->ev[3 + 2 * n_schur], s));
      ++n_schur;
      BA_TRY(launch_solve(h, s));
      BA_TRY(launch_step(h, 1));
    }
    issued += n;
    BA_TRY(hipMemcpyAsync(&h->h_state, h->d_state.p, sizeof(BaState), hipMemcpyDeviceToHost, s));
    BA_TRY(hipStreamSynchronize(s));
    if (h->h_state.status != 0) break;
  }
  if (prof) BA_TRY(hipEventRecord(h->ev[1], s));
  BA_TRY(hipMemcpyAsync(&h->h_state, h->d_state.p, sizeof(BaState), hipMemcpyDeviceToHost, s));
  BA_TRY(hipStreamSynchronize(s));
  if (h->h_state.status == 0) return MVO_HIP_ERR_DEVICE;  // the loop above always runs to a verdict
  const uint32_t N = h->prm.N, L = h->prm.L;
  const int cur = h->h_state.cur;
  h->h_poses.resize(N);
  h->h_X.resize(3 * (size_t)L);
  BA_TRY(hipMemcpyAsync(h->h_poses.data(), h->d_poses.as<double>() + (size_t)cur * 12 * N, sizeof(double) * 12 * N,
                        hipMemcpyDeviceToHost, s));
  BA_TRY(hipMemcpyAsync(h->h_X.data(), h->d_X.as<double>() + (size_t)cur * 3 * L, sizeof(double) * 3 * L,
                        hipMemcpyDeviceToHost, s));
  BA_TRY(hipStreamSynchronize(s));
  h->solved = true;
  mvo_hip_ba_timing& tm = h->timing;
  tm.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  tm.device_seconds = tm.schur_seconds = 0.0;
  tm.observations = h->n_obs;
  tm.schur_bytes = 20ull * h->n_obs + 24ull * L;  // (u, v) + frame word per observation, X per landmark
  tm.schur_flops = h->schur_flops;
  // launches that found the solve finished return at once: count only the linearisations that ran
  const int ran = std::min(n_schur, h->h_state.iteration + (h->h_state.termination == 0 ? 1 : 0));
  tm.schur_launches = 0;
  if (prof) {
    float ms = 0.f;
    BA_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    tm.device_seconds = ms * 1e-3;
    for (int i = 0; i < std::min(ran, kMaxSchurEvents); ++i) {
      BA_TRY(hipEventElapsedTime(&ms, h->ev[2 + 2 * i], h->ev[3 + 2 * i]));
      tm.schur_seconds += ms * 1e-3;
      ++tm.schur_launches;
    }
  }
  return MVO_HIP_OK;
}

int ba_download(mvo_hip_ba_context* h, mvo_hip_ba_pose* poses, double* X, mvo_hip_ba_summary* sm) {
  if (!h->uploaded || !h->solved) return MVO_HIP_ERR_STATE;
  if (poses) std::memcpy(poses, h->h_poses.data(), sizeof(mvo_hip_ba_pose) * h->prm.M);  // only the free frames change
  if (X) std::memcpy(X, h->h_X.data(), sizeof(double) * 3 * (size_t)h->prm.L);
  if (sm) {
    sm->termination = h->h_state.termination;
    sm->iterations = h->h_state.iteration;
    sm->num_successful_steps = h->h_state.nsucc;
    sm->num_unsuccessful_steps = h->h_state.nunsucc;
    sm->initial_cost = h->h_state.initial_cost;
    sm->final_cost = h->h_state.cost;
  }
  return MVO_HIP_OK;
}

}  // namespace

extern "C" {

int mvo_hip_ba_create(int device_ordinal, mvo_hip_ba_handle* out) {
  if (!out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MVO_HIP_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= count) return MVO_HIP_ERR_INVALID_ARGUMENT;
  mvo_hip_ba_context* h = new (std::nothrow) mvo_hip_ba_context();
  if (!h) return MVO_HIP_ERR_OUT_OF_MEMORY;
  h->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return MVO_HIP_ERR_DEVICE;
  }
  *out = h;
  return MVO_HIP_OK;
}

void mvo_hip_ba_destroy(mvo_hip_ba_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  Buf* bufs[] = {&h->d_poses, &h->d_X, &h->d_lm_off, &h->d_lm_nfree, &h->d_obs_frame, &h->d_obs_uv, &h->d_obs_lm,
                 &h->d_obs_src, &h->d_wg_lm, &h->d_wg_lm_step, &h->d_partS, &h->d_partV, &h->d_redS, &h->d_redV, &h->d_partT, &h->d_lmC, &h->d_list_free, &h->d_list_fixed, &h->d_wg_fixed, &h->d_partF,
                 &h->d_delta, &h->d_state, &h->d_trace, &h->d_res, &h->d_sums4};
  for (Buf* b : bufs) b->release();
  if (h->events_ready)
    for (auto& e : h->ev) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int mvo_hip_ba_upload(mvo_hip_ba_handle h, const mvo_hip_ba_pose* poses, uint32_t M, uint32_t N,
                      const double* landmarks, const mvo_hip_ba_observations* observations,
                      const mvo_hip_ba_options* options) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  return ba_upload(h, poses, M, N, landmarks, observations, options);
}

int mvo_hip_ba_solve(mvo_hip_ba_handle h) { return h ? ba_solve(h) : MVO_HIP_ERR_INVALID_ARGUMENT; }

int mvo_hip_ba_download(mvo_hip_ba_handle h, mvo_hip_ba_pose* poses, double* landmarks, mvo_hip_ba_summary* summary) {
  return h ? ba_download(h, poses, landmarks, summary) : MVO_HIP_ERR_INVALID_ARGUMENT;
}

int mvo_hip_ba_refine(mvo_hip_ba_handle h, mvo_hip_ba_pose* poses, uint32_t M, uint32_t N, double* landmarks,
                      const mvo_hip_ba_observations* observations, const mvo_hip_ba_options* options,
                      mvo_hip_ba_summary* summary) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = ba_upload(h, poses, M, N, landmarks, observations, options);
  if (rc != MVO_HIP_OK) return rc;
  rc = ba_solve(h);
  if (rc != MVO_HIP_OK) return rc;
  return ba_download(h, poses, landmarks, summary);
}

int mvo_hip_ba_eval(mvo_hip_ba_handle h, double* residuals, double* cost) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  BA_TRY(hipSetDevice(h->device));
  int rc = ba_reset_point(h);
  if (rc != MVO_HIP_OK) return rc;
  hipStream_t s = h->stream;
  if (residuals && h->n_obs) {
    BA_TRY(h->d_res.reserve(sizeof(double) * 2 * h->n_obs));
    hipLaunchKernelGGL(k_ba_residuals, dim3((uint32_t)((h->n_obs + 255) / 256)), dim3(256), 0, s, h->prm, h->n_obs,
                       h->d_poses.as<double>(), h->d_X.as<double>(), h->d_obs_lm.as<uint32_t>(),
                       h->d_obs_frame.as<uint32_t>(), h->d_obs_uv.as<double2>(), h->d_obs_src.as<uint32_t>(),
                       h->d_res.as<double>());
    BA_TRY(hipGetLastError());
    BA_TRY(hipMemcpyAsync(residuals, h->d_res.p, sizeof(double) * 2 * h->n_obs, hipMemcpyDeviceToHost, s));
  }
  if (cost) {
    BaParams keep = h->prm;
    h->prm.trace_cap = 0;
    hipError_t e = launch_step(h, 0);
    h->prm = keep;
    BA_TRY(e);
    BaState st;
    BA_TRY(hipMemcpyAsync(&st, h->d_state.p, sizeof(BaState), hipMemcpyDeviceToHost, s));
    BA_TRY(hipStreamSynchronize(s));
    *cost = st.cost;
  }
  BA_TRY(hipStreamSynchronize(s));
  return MVO_HIP_OK;
}

int mvo_hip_ba_reduced(mvo_hip_ba_handle h, double radius, double* S, double* rhs, double* camera_gradient,
                       double* gradient_max_norm, double* cost) {
  if (!h || !(radius > 0.0)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  BA_TRY(hipSetDevice(h->device));
  int rc = ba_reset_point(h);
  if (rc != MVO_HIP_OK) return rc;
  BA_TRY(launch_schur(h, radius));
  const uint32_t M = h->prm.M, n = 6 * M;
  std::vector<double> blocks((size_t)h->prm.n_blk * 36), V(18 * (size_t)M + 2);
  BA_TRY(hipMemcpyAsync(blocks.data(), h->d_redS.p, sizeof(double) * blocks.size(), hipMemcpyDeviceToHost, h->stream));
  BA_TRY(hipMemcpyAsync(V.data(), h->d_redV.p, sizeof(double) * V.size(), hipMemcpyDeviceToHost, h->stream));
  BA_TRY(hipStreamSynchronize(h->stream));
  if (S) {
    for (uint32_t f = 0; f < M; ++f)
      for (uint32_t f2 = f; f2 < M; ++f2) {
        const double* b = blocks.data() + (size_t)ba_blk(M, f, f2) * 36;
        for (int p = 0; p < 6; ++p)
          for (int q = 0; q < 6; ++q) {
            S[(size_t)(6 * f + p) * n + 6 * f2 + q] = b[6 * p + q];
            S[(size_t)(6 * f2 + q) * n + 6 * f + p] = b[6 * p + q];
          }
      }
    for (uint32_t i = 0; i < n; ++i)  // the damping k_ba_solve adds when it loads the system
      S[(size_t)i * n + i] += std::fmin(std::fmax(V[12 * M + i], 1e-6), 1e32) / radius;
  }
  double gm = V[18 * M + 1];
  for (uint32_t i = 0; i < n; ++i) {
    if (rhs) rhs[i] = V[i];
    if (camera_gradient) camera_gradient[i] = V[6 * M + i];
    gm = std::fmax(gm, std::fabs(V[6 * M + i]));
  }
  if (gradient_max_norm) *gradient_max_norm = gm;
  if (cost) *cost = V[18 * M];
  return MVO_HIP_OK;
}

int mvo_hip_ba_partial_begin(mvo_hip_ba_handle h) { return h ? ba_partial_begin(h) : MVO_HIP_ERR_INVALID_ARGUMENT; }
int mvo_hip_ba_partial_linearize(mvo_hip_ba_handle h) {
  return h ? ba_partial_linearize(h) : MVO_HIP_ERR_INVALID_ARGUMENT;
}
int mvo_hip_ba_partial_solve(mvo_hip_ba_handle h) { return h ? ba_partial_solve(h) : MVO_HIP_ERR_INVALID_ARGUMENT; }
int mvo_hip_ba_partial_decide(mvo_hip_ba_handle h, int mode, int32_t* done) {
  if (!h || (mode != 0 && mode != 1)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  return ba_partial_decide(h, mode, done);
}

int mvo_hip_ba_partial_exchange(mvo_hip_ba_handle h, int which, void* device_buffer, uint64_t n_doubles,
                                int direction) {
  if (!h || !device_buffer || which < 0 || which > 2) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  const uint64_t nS = (uint64_t)h->prm.n_blk * 36, nVs = 18ull * h->prm.M + 1;
  BA_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  double* buf = static_cast<double*>(device_buffer);
  auto move = [&](double* lib, double* usr, uint64_t n) {
    return direction == 0 ? hipMemcpyAsync(usr, lib, sizeof(double) * n, hipMemcpyDeviceToDevice, s)
                          : hipMemcpyAsync(lib, usr, sizeof(double) * n, hipMemcpyDeviceToDevice, s);
  };
  if (which == 0) {  // reduced camera system blocks, then rhs | g_c | diag(B) | cost: sums
    if (n_doubles != nS + nVs) return MVO_HIP_ERR_INVALID_ARGUMENT;
    BA_TRY(move(h->d_redS.as<double>(), buf, nS));
    BA_TRY(move(h->d_redV.as<double>(), buf + nS, nVs));
  } else if (which == 1) {  // largest landmark gradient entry: a maximum
    if (n_doubles != 1) return MVO_HIP_ERR_INVALID_ARGUMENT;
    BA_TRY(move(h->d_redV.as<double>() + nVs, buf, 1));
  } else {  // cost | model-cost-change part | |dl|^2 | |X|^2: sums
    if (n_doubles != 4 || !h->d_sums4.p) return MVO_HIP_ERR_INVALID_ARGUMENT;
    BA_TRY(move(h->d_sums4.as<double>(), buf, 4));
  }
  BA_TRY(hipStreamSynchronize(s));
  return MVO_HIP_OK;
}

#ifdef MVO_BA_STAMPS
int mvo_hip_ba_debug_stamps(unsigned long long* out8, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(mvo_ba_stamps), z, sizeof z) == hipSuccess ? 0 : 4;
  }
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(mvo_ba_stamps), 128) == hipSuccess ? 0 : 4;
}
#endif

int mvo_hip_ba_set_trace(mvo_hip_ba_handle h, int32_t capacity) {
  if (!h || capacity < 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  h->trace_cap = capacity;
  return MVO_HIP_OK;
}

int mvo_hip_ba_get_trace(mvo_hip_ba_handle h, mvo_hip_ba_iteration* rows, int32_t capacity, int32_t* n_rows) {
  if (!h || !n_rows) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->solved) return MVO_HIP_ERR_STATE;
  *n_rows = h->h_state.ntrace;
  int32_t n = std::min(h->h_state.ntrace, h->trace_cap);
  n = std::min(n, capacity);
  if (rows && n > 0) {
    BA_TRY(hipSetDevice(h->device));
    BA_TRY(hipMemcpy(rows, h->d_trace.p, sizeof(mvo_hip_ba_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  return MVO_HIP_OK;
}

int mvo_hip_ba_set_profiling(mvo_hip_ba_handle h, int enabled) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  h->profiling = enabled != 0;
  return MVO_HIP_OK;
}

int mvo_hip_ba_get_timing(mvo_hip_ba_handle h, mvo_hip_ba_timing* out) {
  if (!h || !out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = h->timing;
  return MVO_HIP_OK;
}

}  // extern "C"

// mvo_kernels.h -- hand-written gfx950 kernels of libmvo_hip (included once by mvo_hip.hip).
//
//   k_fold   observation pass #0: c = (R_a m0) x (R_{b+1} m1) per residual block, optional
//            store of c (SoA planes) and/or per-tile second moments S = sum c c^T.
//   k_pass   evaluation pass: r = u_p . c, Cauchy weight, weighted moments + sum rho.
//   k_lm     one workgroup per window: Ceres-rule Levenberg-Marquardt on the <= 15 x 15
//            normal equations assembled from the per-pair moments.
//   k_eval   debug: per-residual-block r and local Jacobian row.
//
// Math: r_i = u_p . c_i with the per-pair unit baseline u_p, so the
// Jacobian row of block i is c_i^T A_p with a per-pair 3 x 3|F_p| matrix A_p, and
//   J^T J = sum_p A_p^T S_p A_p,  J^T r = sum_p A_p^T S_p u_p,  S_p = sum_i w_i c_i c_i^T.
// All reductions are wavefront (64-lane) shuffles -> LDS -> one partial per workgroup slot;
// no global atomics, fixed summation order (bitwise reproducible).
#pragma once

#include "mvo_device.h"

#define FOLD_THREADS 256
#define FOLD_WAVES 4

// ---------------------------------------------------------------------------------------
// workgroup epilogue shared by k_fold / k_pass: red[w][8] holds each wave's tile sums; the
// waves of a workgroup whose tiles share a slot are summed in wave order and stored once.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void block_store_partials(const TileDev* __restrict__ tiles, uint32_t n_tiles,
                                                     double (*red)[8], double* __restrict__ partial) {
  __syncthreads();
  const int v = threadIdx.x;
  if (v < 8) {
    double acc = 0.0;
    const uint32_t t0 = blockIdx.x * FOLD_WAVES;
#pragma unroll
    for (int w = 0; w < FOLD_WAVES; ++w) {
      const uint32_t t = t0 + w;
      if (t >= n_tiles) break;
      acc += red[w][v];
      const uint32_t slot = tiles[t].slot;
      const bool last = (w == FOLD_WAVES - 1) || (t + 1 >= n_tiles) || (tiles[t + 1].slot != slot);
      if (last) {
        partial[(size_t)slot * 8 + v] = acc;
        acc = 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_fold: reads m0, m1 (48 B / residual block, caller's interleaved layout), optionally writes
// c (24 B / block) and/or accumulates S.  One wavefront per tile.
// ---------------------------------------------------------------------------------------
template <bool WRITE_C, bool MOMENTS>
__global__ __launch_bounds__(FOLD_THREADS) void k_fold(const TileDev* __restrict__ tiles, uint32_t n_tiles,
                                                       const double* __restrict__ frames,
                                                       const double* __restrict__ m0,
                                                       const double* __restrict__ m1,
                                                       double* __restrict__ cbuf, size_t cstride,
                                                       double* __restrict__ partial) {
  __shared__ double red[FOLD_WAVES][8];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * FOLD_WAVES + wv;
  double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0;
  if (t < n_tiles) {
    const TileDev tl = tiles[t];
    const double* __restrict__ R0 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f0;
    const double* __restrict__ R1 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f1;
    const double a0 = R0[0], a1 = R0[1], a2 = R0[2], a3 = R0[3], a4 = R0[4], a5 = R0[5], a6 = R0[6],
                 a7 = R0[7], a8 = R0[8];
    const double b0 = R1[0], b1 = R1[1], b2 = R1[2], b3 = R1[3], b4 = R1[4], b5 = R1[5], b6 = R1[6],
                 b7 = R1[7], b8 = R1[8];
    const double* __restrict__ p0 = m0 + 3 * (size_t)tl.src_start;
    const double* __restrict__ p1 = m1 + 3 * (size_t)tl.src_start;
    double* __restrict__ cx = cbuf + tl.c_start;
    double* __restrict__ cy = cx + cstride;
    double* __restrict__ cz = cy + cstride;
#pragma unroll 4
    for (uint32_t i = lane; i < tl.count; i += 64) {
      const double x0 = p0[3 * i], y0 = p0[3 * i + 1], z0 = p0[3 * i + 2];
      const double x1 = p1[3 * i], y1 = p1[3 * i + 1], z1 = p1[3 * i + 2];
      const double px = a0 * x0 + a1 * y0 + a2 * z0;  // R0 * m0
      const double py = a3 * x0 + a4 * y0 + a5 * z0;
      const double pz = a6 * x0 + a7 * y0 + a8 * z0;
      const double qx = b0 * x1 + b1 * y1 + b2 * z1;  // R1 * m1
      const double qy = b3 * x1 + b4 * y1 + b5 * z1;
      const double qz = b6 * x1 + b7 * y1 + b8 * z1;
      const double vx = py * qz - pz * qy;  // c = p x q
      const double vy = pz * qx - px * qz;
      const double vz = px * qy - py * qx;
      if (WRITE_C) {
        cx[i] = vx;
        cy[i] = vy;
        cz[i] = vz;
      }
      if (MOMENTS) {
        S0 += vx * vx;
        S1 += vx * vy;
        S2 += vx * vz;
        S3 += vy * vy;
        S4 += vy * vz;
        S5 += vz * vz;
      }
    }
  }
  if (MOMENTS) {
    S0 = wave_sum(S0);
    S1 = wave_sum(S1);
    S2 = wave_sum(S2);
    S3 = wave_sum(S3);
    S4 = wave_sum(S4);
    S5 = wave_sum(S5);
    if (lane == 0) {
      red[wv][0] = S0;
      red[wv][1] = S1;
      red[wv][2] = S2;
      red[wv][3] = S3;
      red[wv][4] = S4;
      red[wv][5] = S5;
      red[wv][6] = 0.0;
      red[wv][7] = 0.0;
    }
    block_store_partials(tiles, n_tiles, red, partial);
  }
}

// ---------------------------------------------------------------------------------------
// k_pass: one evaluation of all residual blocks at the unit baselines in pair_u.
// reads c (24 B / block) as 16-byte loads from three SoA planes.
// partial[slot] = { S (6, weighted), sum rho(r^2) or sum r^2, sum w r^2 }
// ---------------------------------------------------------------------------------------
template <bool ROBUST>
__device__ __forceinline__ void pass_accum(double cx, double cy, double cz, double u0, double u1, double u2,
                                           double& S0, double& S1, double& S2, double& S3, double& S4,
                                           double& S5, double& rho, double& wr2) {
  const double r = u0 * cx + u1 * cy + u2 * cz;
  const double s = r * r;
  double w = 1.0;
  if (ROBUST) {
    // ceres::CauchyLoss(0.5): b = 0.25, c = 4; rho = b log(1 + s c); rho' = 1 / (1 + s c)
    const double sum = 1.0 + s * 4.0;
    const double inv = 1.0 / sum;
    rho += 0.25 * log(sum);
    w = fmax(DBL_MIN, inv);
  } else {
    rho += s;
  }
  wr2 += w * s;
  const double wx = w * cx, wy = w * cy, wz = w * cz;
  S0 += wx * cx;
  S1 += wx * cy;
  S2 += wx * cz;
  S3 += wy * cy;
  S4 += wy * cz;
  S5 += wz * cz;
}

template <bool ROBUST>
__global__ __launch_bounds__(FOLD_THREADS) void k_pass(const TileDev* __restrict__ tiles, uint32_t n_tiles,
                                                       const double* __restrict__ cbuf, size_t cstride,
                                                       const double* __restrict__ pair_u,
                                                       const int32_t* __restrict__ win_status,
                                                       double* __restrict__ partial) {
  __shared__ double red[FOLD_WAVES][8];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * FOLD_WAVES + wv;
  double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0, rho = 0, wr2 = 0;
  if (t < n_tiles) {
    const TileDev tl = tiles[t];
    if (win_status[tl.win] == 0) {
      const double u0 = pair_u[4 * (size_t)tl.pair], u1 = pair_u[4 * (size_t)tl.pair + 1],
                   u2 = pair_u[4 * (size_t)tl.pair + 2];
      const double* __restrict__ cx = cbuf + tl.c_start;
      const double* __restrict__ cy = cx + cstride;
      const double* __restrict__ cz = cy + cstride;
#pragma unroll 2
      for (uint32_t i = 2 * lane; i < tl.count; i += 128) {
        const double2 x = *reinterpret_cast<const double2*>(cx + i);
        const double2 y = *reinterpret_cast<const double2*>(cy + i);
        const double2 z = *reinterpret_cast<const double2*>(cz + i);
        pass_accum<ROBUST>(x.x, y.x, z.x, u0, u1, u2, S0, S1, S2, S3, S4, S5, rho, wr2);
        const bool v1 = (i + 1) < tl.count;  // the pad element of an odd tile is not data
        pass_accum<ROBUST>(v1 ? x.y : 0.0, v1 ? y.y : 0.0, v1 ? z.y : 0.0, u0, u1, u2, S0, S1, S2, S3,
                           S4, S5, rho, wr2);
      }
    }
  }
  S0 = wave_sum(S0);
  S1 = wave_sum(S1);
  S2 = wave_sum(S2);
  S3 = wave_sum(S3);
  S4 = wave_sum(S4);
  S5 = wave_sum(S5);
  rho = wave_sum(rho);
  wr2 = wave_sum(wr2);
  if (lane == 0) {
    red[wv][0] = S0;
    red[wv][1] = S1;
    red[wv][2] = S2;
    red[wv][3] = S3;
    red[wv][4] = S4;
    red[wv][5] = S5;
    red[wv][6] = rho;
    red[wv][7] = wr2;
  }
  block_store_partials(tiles, n_tiles, red, partial);
}

// ---------------------------------------------------------------------------------------
// k_lm
// ---------------------------------------------------------------------------------------
struct LmShared {
  LmState st;
  double zvec[MVO_MAXM][3], zt[MVO_MAXM], zs[MVO_MAXM], zsp[MVO_MAXM];
  double G[MVO_MAXM][9];  // G[k][3*row + col]: columns (s_k jl0, s_k jl1, s'_k v_k)
  double cls[MVO_MAXM * (MVO_MAXM + 1)][10];  // class (a, j = min(b, M))
  double Hz[MVO_MAXC * MVO_MAXC], gz[MVO_MAXC];  // full 3M system at z
  double A[MVO_MAXC * MVO_MAXC], rhs[MVO_MAXC], step[MVO_MAXC];
  double cost_z;
  int32_t act[MVO_MAXC];  // active column -> full column
  int32_t n, go;
};

// compact column index of full column 3k+j (valid only for active columns)
__device__ __forceinline__ int ccol(int k, int j, int fix) { return fix ? 2 * k + j : 3 * k + j; }

__device__ __forceinline__ void lm_point_plus(const LmShared& sh, int M, int fix, const double (*xv)[3],
                                              const double* xt, const double* delta, double sign,
                                              double (*ov)[3], double* ot) {
  for (int k = 0; k < M; ++k) {
    d_plus(xv[k], sign * delta[ccol(k, 0, fix)], sign * delta[ccol(k, 1, fix)], ov[k]);
    ot[k] = fix ? xt[k] : xt[k] + sign * delta[ccol(k, 2, fix)];
  }
  (void)sh;
}

__device__ __forceinline__ double lm_diff_norm(int M, int fix, const double (*av)[3], const double* at,
                                               const double (*bv)[3], const double* bt, bool inf_norm) {
  double acc = 0.0;
  for (int k = 0; k < M; ++k) {
    for (int j = 0; j < 3; ++j) {
      const double d = fabs(av[k][j] - (bv ? bv[k][j] : 0.0));
      acc = inf_norm ? fmax(acc, d) : acc + d * d;
    }
    if (!fix) {
      const double d = fabs(at[k] - (bt ? bt[k] : 0.0));
      acc = inf_norm ? fmax(acc, d) : acc + d * d;
    }
  }
  return inf_norm ? acc : sqrt(acc);
}

__device__ __forceinline__ void lm_terminate(LmState& st, int termination, int reason) {
  st.status = 1;
  st.termination = termination;
  st.reason = reason;
}

// Thread-0 control logic: Ceres 1.14 TrustRegionMinimizer::Minimize restated around the
// evaluation that was just completed at z (= x0 in PH_EVAL_X0, the candidate in PH_EVAL_CAND).
// Returns 1 when a new candidate awaits evaluation (in sh.zvec/zt and st.c_*), 0 when done.
__device__ int lm_advance(LmShared& sh, const SolveParams& prm, int M, TraceRec* __restrict__ trace) {
  LmState& st = sh.st;
  const int n = sh.n, fix = prm.fix_length;
  TraceRec it;
  it.cost = it.cost_change = it.gradient_max_norm = it.step_norm = it.relative_decrease = 0.0;
  it.trust_region_radius = 0.0;
  it.step_is_valid = it.step_is_successful = 0;

  bool take_z = false;
  if (st.phase == PH_EVAL_X0) {
    // IterationZero
    take_z = true;
    st.iteration = 0;
    st.min_cost = DBL_MAX;
    st.radius = 1e4;  // initial_trust_region_radius
    st.decrease_factor = 2.0;
    st.reuse_diagonal = 0;
    st.num_invalid = 0;
    st.nsucc = st.nunsucc = 0;
    st.ntrace = 0;
    st.model_cost_change = 0.0;
  } else {
    const double cand_cost = sh.cost_z;
    // ParameterToleranceReached
    it.step_norm = lm_diff_norm(M, fix, st.x_vec, st.x_t, st.c_vec, st.c_t, false);
    if (it.step_norm <= prm.ptol * (st.x_norm + prm.ptol)) {
      lm_terminate(st, 0, 3);
      return 0;
    }
    // FunctionToleranceReached
    it.cost_change = st.x_cost - cand_cost;
    if (fabs(it.cost_change) <= prm.ftol * st.x_cost) {
      lm_terminate(st, 0, 4);
      return 0;
    }
    // TrustRegionStepEvaluator::StepQuality
    const double rel = (st.se_cur - cand_cost) / st.model_cost_change;
    const double hist = (st.se_ref - cand_cost) / (st.se_acc_ref + st.model_cost_change);
    it.relative_decrease = fmax(rel, hist);
    it.step_is_valid = 1;
    if (it.relative_decrease > 1e-3) {  // min_relative_decrease
      take_z = true;
      it.step_is_successful = 1;
      // LevenbergMarquardtStrategy::StepAccepted
      const double q = 2.0 * it.relative_decrease - 1.0;
      st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
      st.radius = fmin(1e16, st.radius);
      st.decrease_factor = 2.0;
      st.reuse_diagonal = 0;
      // TrustRegionStepEvaluator::StepAccepted
      st.se_cur = cand_cost;
      st.se_acc_cand += st.model_cost_change;
      st.se_acc_ref += st.model_cost_change;
      if (st.se_cur < st.se_min) {
        st.se_min = st.se_cur;
        st.se_nonmono = 0;
        st.se_cand = st.se_cur;
        st.se_acc_cand = 0.0;
      } else {
        ++st.se_nonmono;
        if (st.se_cur > st.se_cand) {
          st.se_cand = st.se_cur;
          st.se_acc_cand = 0.0;
        }
      }
      if (st.se_nonmono == 5) {  // max_consecutive_nonmonotonic_steps
        st.se_ref = st.se_cand;
        st.se_acc_ref = st.se_acc_cand;
      }
    } else {
      // HandleUnsuccessfulStep + StepRejected
      it.step_is_successful = 0;
      st.radius = st.radius / st.decrease_factor;
      st.decrease_factor *= 2.0;
      st.reuse_diagonal = 1;
      it.cost = cand_cost;
      it.gradient_max_norm = st.grad_max_norm;
    }
  }

  if (take_z) {
    // x <- z; EvaluateGradientAndJacobian results = the system assembled at z
    for (int k = 0; k < M; ++k) {
      for (int j = 0; j < 3; ++j) st.x_vec[k][j] = sh.zvec[k][j];
      st.x_t[k] = sh.zt[k];
    }
    st.x_cost = sh.cost_z;
    st.x_norm = lm_diff_norm(M, fix, st.x_vec, st.x_t, nullptr, nullptr, false);
    for (int i = 0; i < n; ++i) {
      st.g[i] = sh.gz[sh.act[i]];
      for (int j = 0; j < n; ++j) st.H[i * MVO_MAXC + j] = sh.Hz[sh.act[i] * MVO_MAXC + sh.act[j]];
    }
    if (st.phase == PH_EVAL_X0) {
      // jacobi_scaling, fixed at iteration 0: 1 / (1 + ||J_col||)
      for (int j = 0; j < n; ++j) st.jscale[j] = 1.0 / (1.0 + sqrt(st.H[j * MVO_MAXC + j]));
      st.initial_cost = st.final_cost = st.x_cost;
      st.se_min = st.se_cur = st.se_ref = st.se_cand = st.x_cost;
      st.se_acc_ref = st.se_acc_cand = 0.0;
      st.se_nonmono = 0;
      it.step_is_valid = 1;
      it.step_is_successful = 1;
    }
    // gradient_max_norm = || x - Plus(x, -g) ||_inf
    {
      double pv[MVO_MAXM][3], pt[MVO_MAXM];
      lm_point_plus(sh, M, fix, st.x_vec, st.x_t, st.g, -1.0, pv, pt);
      st.grad_max_norm = lm_diff_norm(M, fix, st.x_vec, st.x_t, pv, pt, true);
    }
    it.cost = st.x_cost;
    it.gradient_max_norm = st.grad_max_norm;
  }

  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
    if (it.step_is_successful) {
      ++st.nsucc;
      if (st.x_cost < st.min_cost) {
        st.min_cost = st.x_cost;
        for (int k = 0; k < M; ++k) {
          for (int j = 0; j < 3; ++j) st.b_vec[k][j] = st.x_vec[k][j];
          st.b_t[k] = st.x_t[k];
        }
      }
    } else {
      ++st.nunsucc;
    }
    it.trust_region_radius = st.radius;
    if (trace && st.ntrace < prm.trace_cap) trace[st.ntrace] = it;
    ++st.ntrace;
    if (it.cost < st.final_cost) st.final_cost = it.cost;

    if (st.iteration >= prm.max_it) {
      lm_terminate(st, 1, 1);
      return 0;
    }
    if (it.step_is_successful && it.gradient_max_norm <= prm.gtol) {
      lm_terminate(st, 0, 2);
      return 0;
    }
    if (st.radius <= 1e-32) {  // min_trust_region_radius
      lm_terminate(st, 0, 5);
      return 0;
    }

    // ---- next iteration: ComputeTrustRegionStep ----
    ++st.iteration;
    it.cost = it.cost_change = it.gradient_max_norm = it.step_norm = it.relative_decrease = 0.0;
    it.step_is_valid = it.step_is_successful = 0;

    if (!st.reuse_diagonal) {
      for (int j = 0; j < n; ++j) {
        const double d = st.jscale[j] * st.jscale[j] * st.H[j * MVO_MAXC + j];
        st.diag[j] = fmin(fmax(d, 1e-6), 1e32);  // min/max_lm_diagonal
      }
    }
    st.reuse_diagonal = 1;
    // (Js^T Js + D^2) y = Js^T r,  Js = J diag(jscale),  D^2 = diag / radius; step = -y.
    // Ceres solves the equivalent least-squares problem by dense QR; the system here is at
    // most 15 x 15 and already reduced, so it is factorised by Cholesky in LDS.
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j)
        sh.A[i * MVO_MAXC + j] = st.jscale[i] * st.H[i * MVO_MAXC + j] * st.jscale[j];
      sh.A[i * MVO_MAXC + i] += st.diag[i] / st.radius;
      sh.rhs[i] = st.jscale[i] * st.g[i];
    }
    bool ok = true;
    for (int j = 0; j < n && ok; ++j) {
      double d = sh.A[j * MVO_MAXC + j];
      for (int k = 0; k < j; ++k) d -= sh.A[j * MVO_MAXC + k] * sh.A[j * MVO_MAXC + k];
      if (!(d > 0.0) || isinf(d)) {
        ok = false;
        break;
      }
      const double l = sqrt(d);
      sh.A[j * MVO_MAXC + j] = l;
      for (int i = j + 1; i < n; ++i) {
        double s = sh.A[i * MVO_MAXC + j];
        for (int k = 0; k < j; ++k) s -= sh.A[i * MVO_MAXC + k] * sh.A[j * MVO_MAXC + k];
        sh.A[i * MVO_MAXC + j] = s / l;
      }
    }
    double mcc = 0.0;
    if (ok) {
      for (int i = 0; i < n; ++i) {  // L w = rhs
        double s = sh.rhs[i];
        for (int k = 0; k < i; ++k) s -= sh.A[i * MVO_MAXC + k] * sh.step[k];
        sh.step[i] = s / sh.A[i * MVO_MAXC + i];
      }
      for (int i = n - 1; i >= 0; --i) {  // L^T y = w
        double s = sh.step[i];
        for (int k = i + 1; k < n; ++k) s -= sh.A[k * MVO_MAXC + i] * sh.step[k];
        sh.step[i] = s / sh.A[i * MVO_MAXC + i];
      }
      for (int i = 0; i < n; ++i) {
        sh.step[i] = -sh.step[i];
        if (!isfinite(sh.step[i])) ok = false;
      }
    }
    if (ok) {
      // model_cost_change = -(Js step)^T (r + Js step / 2) = -step^T Js^T r - step^T Js^T Js step / 2
      double lin = 0.0, quad = 0.0;
      for (int i = 0; i < n; ++i) {
        lin += sh.step[i] * st.jscale[i] * st.g[i];
        double row = 0.0;
        for (int j = 0; j < n; ++j) row += st.jscale[i] * st.H[i * MVO_MAXC + j] * st.jscale[j] * sh.step[j];
        quad += sh.step[i] * row;
      }
      mcc = -(lin + 0.5 * quad);
    }
    st.model_cost_change = mcc;
    const bool valid = ok && mcc > 0.0;
    if (!valid) {
      // ---- HandleInvalidStep ----
      if (++st.num_invalid >= 5) {  // max_num_consecutive_invalid_steps
        lm_terminate(st, 2, 6);
        return 0;
      }
      st.radius *= 0.5;  // StepIsInvalid
      st.reuse_diagonal = 1;
      it.cost = st.x_cost;
      it.gradient_max_norm = st.grad_max_norm;
      continue;
    }
    st.num_invalid = 0;
    for (int i = 0; i < n; ++i) st.delta[i] = sh.step[i] * st.jscale[i];
    // ---- ComputeCandidatePoint ----
    lm_point_plus(sh, M, fix, st.x_vec, st.x_t, st.delta, 1.0, st.c_vec, st.c_t);
    for (int k = 0; k < M; ++k) {
      for (int j = 0; j < 3; ++j) sh.zvec[k][j] = st.c_vec[k][j];
      sh.zt[k] = st.c_t[k];
    }
    st.phase = PH_EVAL_CAND;
    return 1;
  }
}

// per-pair index of (a, b) in the reference's loop order
__device__ __forceinline__ int pair_index(int N, int a, int b) { return a * (N - 1) - (a * (a - 1)) / 2 + (b - a); }

template <int NT>
__global__ __launch_bounds__(NT) void k_lm(SolveParams prm, const WinDev* __restrict__ wins,
                                           const double* __restrict__ frames, const double* __restrict__ init_x,
                                           const PairDev* __restrict__ pairs, const double* __restrict__ partial,
                                           LmState* __restrict__ states, LmResult* __restrict__ results,
                                           double* __restrict__ pair_u, int32_t* __restrict__ win_status,
                                           TraceRec* __restrict__ trace, int32_t* __restrict__ done_count) {
  extern __shared__ double pairbuf[];  // P x 10: T(6) q(3) cost(1)
  __shared__ LmShared sh;
  const int w = blockIdx.x, tid = threadIdx.x;
  LmState* __restrict__ gst = states + w;
  if (gst->status != 0) return;  // uniform: every thread reads the same word
  const WinDev win = wins[w];
  const int M = win.M, N = win.N, P = win.P;
  const int fix = prm.fix_length;
  const double* __restrict__ fr = frames + (size_t)MVO_FRAME_DOUBLES * win.frame_off;
  const PairDev* __restrict__ wp = pairs + win.pair_off;

  for (int i = tid; i < (int)(sizeof(LmState) / 8); i += NT)
    reinterpret_cast<double*>(&sh.st)[i] = reinterpret_cast<const double*>(gst)[i];
  __syncthreads();
  LmState& st = sh.st;

  if (tid == 0) {
    int n = 0;
    for (int k = 0; k < M; ++k) {
      sh.act[n++] = 3 * k;
      sh.act[n++] = 3 * k + 1;
      if (!fix) sh.act[n++] = 3 * k + 2;
    }
    sh.n = n;
    if (st.phase == PH_START) {
      // IR.cpp:95-109: vec_k from the frame, t_k = reverseScale(scale_k) (host libm, uploaded)
      for (int k = 0; k < M; ++k) {
        for (int j = 0; j < 3; ++j) st.x_vec[k][j] = init_x[((size_t)w * MVO_MAXM + k) * 4 + j];
        st.x_t[k] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 3];
      }
    }
    const bool at_x = (st.phase != PH_EVAL_CAND);
    for (int k = 0; k < M; ++k) {
      for (int j = 0; j < 3; ++j) sh.zvec[k][j] = at_x ? st.x_vec[k][j] : st.c_vec[k][j];
      sh.zt[k] = at_x ? st.x_t[k] : st.c_t[k];
    }
  }
  __syncthreads();

  bool need_pass = false;
  if (st.phase == PH_START) {
    __syncthreads();
    if (tid == 0) st.phase = PH_EVAL_X0;
    __syncthreads();
    need_pass = prm.pass_mode != 0;
  }

  while (!need_pass) {
    // ---------------- evaluation at z: assemble J^T J, J^T r, cost ----------------
    if (tid < M) {
      const int k = tid;
      const double s = d_scale(sh.zt[k], prm.max_len, prm.min_len);
      const double sp = d_dscale(sh.zt[k], prm.max_len, prm.min_len);
      sh.zs[k] = s;
      sh.zsp[k] = sp;
      const double vx = sh.zvec[k][0], vy = sh.zvec[k][1], vz = sh.zvec[k][2];
      // J_loc(v) = [[-2vy, 2vz], [2vx, 0], [0, -2vx]]  (d Plus / d delta at 0)
      sh.G[k][0] = s * (-2.0 * vy); sh.G[k][1] = s * (2.0 * vz);  sh.G[k][2] = sp * vx;
      sh.G[k][3] = s * (2.0 * vx);  sh.G[k][4] = 0.0;             sh.G[k][5] = sp * vy;
      sh.G[k][6] = 0.0;             sh.G[k][7] = s * (-2.0 * vx); sh.G[k][8] = sp * vz;
    }
    __syncthreads();
    for (int p = tid; p < P; p += NT) {
      const PairDev pd = wp[p];
      double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t sl = pd.slot_begin; sl < pd.slot_end; ++sl) {
        const double* __restrict__ ps = partial + (size_t)sl * 8;
#pragma unroll
        for (int v = 0; v < 8; ++v) S[v] += ps[v];
      }
      double u[3], nb;
      pair_unit_baseline(fr, M, (int)pd.a, (int)pd.b, sh.zvec, sh.zs, u, nb);
      const double Su0 = S[0] * u[0] + S[1] * u[1] + S[2] * u[2];
      const double Su1 = S[1] * u[0] + S[3] * u[1] + S[4] * u[2];
      const double Su2 = S[2] * u[0] + S[4] * u[1] + S[5] * u[2];
      const double uSu = u[0] * Su0 + u[1] * Su1 + u[2] * Su2;
      const double inb = 1.0 / nb, inb2 = inb * inb;
      double* __restrict__ o = pairbuf + (size_t)p * 10;
      // T = P S P, P = (I - u u^T)/nb
      o[0] = (S[0] - 2.0 * u[0] * Su0 + u[0] * u[0] * uSu) * inb2;
      o[1] = (S[1] - u[0] * Su1 - Su0 * u[1] + u[0] * u[1] * uSu) * inb2;
      o[2] = (S[2] - u[0] * Su2 - Su0 * u[2] + u[0] * u[2] * uSu) * inb2;
      o[3] = (S[3] - 2.0 * u[1] * Su1 + u[1] * u[1] * uSu) * inb2;
      o[4] = (S[4] - u[1] * Su2 - Su1 * u[2] + u[1] * u[2] * uSu) * inb2;
      o[5] = (S[5] - 2.0 * u[2] * Su2 + u[2] * u[2] * uSu) * inb2;
      // q = P S u
      o[6] = (Su0 - u[0] * uSu) * inb;
      o[7] = (Su1 - u[1] * uSu) * inb;
      o[8] = (Su2 - u[2] * uSu) * inb;
      // cost: moment form 1/2 u^T S u, or 1/2 sum rho from the pass
      o[9] = prm.pass_mode ? 0.5 * S[6] : 0.5 * uSu;
    }
    __syncthreads();
    // Class sums in pair order.  Class (a, j), j = min(b, M): j < M is the single pair b = j,
    // j = M collects the pairs b = M .. N-2 (those with a fixed-frame offset).  The free frames
    // of a class are a .. min(j, M-1).
    const int MC = M + 1;
    for (int item = tid; item < M * MC * 10; item += NT) {
      const int a = item / (MC * 10), j = (item / 10) % MC, v = item % 10;
      double acc = 0.0;
      if (j >= a) {
        const int b0 = j, b1 = (j == M) ? N - 2 : j;
        if (b0 <= N - 2) {
          const int p0 = pair_index(N, a, b0);
          for (int b = b0; b <= b1; ++b) acc += pairbuf[(size_t)(p0 + (b - b0)) * 10 + v];
        }
      }
      sh.cls[a * MC + j][v] = acc;
    }
    __syncthreads();
    for (int item = tid; item < M * M + M + 1; item += NT) {
      if (item < M * M) {
        const int k = item / M, l = item % M;
        if (l >= k) {
          // W = sum of T over the pairs whose free frames contain k and l: a <= k, min(j,M-1) >= l.
          // The pair (k,k) is kept apart: its baseline is s_k v_k alone, so r does not depend on
          // t_k at all (IR.cpp:466-471 normalises the scale away) and its scale row/column is
          // exactly zero instead of rounding noise of the size of S.
          double W[6] = {0, 0, 0, 0, 0, 0};
          for (int a = 0; a <= k; ++a)
            for (int j = l; j <= M; ++j) {
              if (k == l && a == k && j == k) continue;
              for (int v = 0; v < 6; ++v) W[v] += sh.cls[a * MC + j][v];
            }
          double blk[9];
          {
            const double Wm[9] = {W[0], W[1], W[2], W[1], W[3], W[4], W[2], W[4], W[5]};
            double WG[9];
            for (int r = 0; r < 3; ++r)
              for (int c = 0; c < 3; ++c)
                WG[3 * r + c] = Wm[3 * r] * sh.G[l][c] + Wm[3 * r + 1] * sh.G[l][3 + c] + Wm[3 * r + 2] * sh.G[l][6 + c];
            for (int i = 0; i < 3; ++i)
              for (int jj = 0; jj < 3; ++jj)
                blk[3 * i + jj] = sh.G[k][i] * WG[jj] + sh.G[k][3 + i] * WG[3 + jj] + sh.G[k][6 + i] * WG[6 + jj];
          }
          if (k == l) {
            const double* Ts = sh.cls[k * MC + k];
            const double Wm[9] = {Ts[0], Ts[1], Ts[2], Ts[1], Ts[3], Ts[4], Ts[2], Ts[4], Ts[5]};
            double WG[9];
            for (int r = 0; r < 3; ++r)
              for (int c = 0; c < 3; ++c)
                WG[3 * r + c] = Wm[3 * r] * sh.G[k][c] + Wm[3 * r + 1] * sh.G[k][3 + c] + Wm[3 * r + 2] * sh.G[k][6 + c];
            for (int i = 0; i < 2; ++i)
              for (int jj = 0; jj < 2; ++jj)
                blk[3 * i + jj] += sh.G[k][i] * WG[jj] + sh.G[k][3 + i] * WG[3 + jj] + sh.G[k][6 + i] * WG[6 + jj];
          }
          for (int i = 0; i < 3; ++i)
            for (int jj = 0; jj < 3; ++jj) {
              sh.Hz[(3 * k + i) * MVO_MAXC + 3 * l + jj] = blk[3 * i + jj];
              sh.Hz[(3 * l + jj) * MVO_MAXC + 3 * k + i] = blk[3 * i + jj];
            }
        }
      } else if (item < M * M + M) {
        const int k = item - M * M;
        double y[3] = {0, 0, 0};
        for (int a = 0; a <= k; ++a)
          for (int j = k; j <= M; ++j) {
            if (a == k && j == k) continue;
            for (int v = 0; v < 3; ++v) y[v] += sh.cls[a * MC + j][6 + v];
          }
        const double* qs = &sh.cls[k * MC + k][6];
        for (int i = 0; i < 3; ++i) {
          double gi = sh.G[k][i] * y[0] + sh.G[k][3 + i] * y[1] + sh.G[k][6 + i] * y[2];
          if (i < 2) gi += sh.G[k][i] * qs[0] + sh.G[k][3 + i] * qs[1] + sh.G[k][6 + i] * qs[2];
          sh.gz[3 * k + i] = gi;
        }
      } else {
        double c = 0.0;
        for (int a = 0; a < M; ++a)
          for (int j = a; j <= M; ++j) c += sh.cls[a * MC + j][9];
        sh.cost_z = c;
      }
    }
    __syncthreads();
    // ---------------- thread 0: accept/reject, damping, next candidate ----------------
    if (tid == 0) sh.go = lm_advance(sh, prm, M, (w == 0) ? trace : nullptr);
    __syncthreads();
    if (!sh.go) break;
    if (prm.pass_mode) need_pass = true;
  }

  if (need_pass) {
    // publish the unit baselines of the point to evaluate; k_pass runs next on the stream
    if (tid < M) sh.zs[tid] = d_scale(sh.zt[tid], prm.max_len, prm.min_len);
    __syncthreads();
    for (int p = tid; p < P; p += NT) {
      const PairDev pd = wp[p];
      double u[3], nb;
      pair_unit_baseline(fr, M, (int)pd.a, (int)pd.b, sh.zvec, sh.zs, u, nb);
      double* __restrict__ o = pair_u + 4 * (size_t)(win.pair_off + p);
      o[0] = u[0];
      o[1] = u[1];
      o[2] = u[2];
      o[3] = nb;
    }
  }
  __syncthreads();
  for (int i = tid; i < (int)(sizeof(LmState) / 8); i += NT)
    reinterpret_cast<double*>(gst)[i] = reinterpret_cast<const double*>(&sh.st)[i];
  if (tid == 0 && st.status != 0) {
    LmResult r;
    for (int k = 0; k < MVO_MAXM; ++k) {
      for (int j = 0; j < 3; ++j) r.vec[k][j] = st.b_vec[k][j];
      r.t[k] = st.b_t[k];
    }
    r.initial_cost = st.initial_cost;
    r.final_cost = st.final_cost;
    r.termination = st.termination;
    r.reason = st.reason;
    r.iterations = st.ntrace - 1;
    r.nsucc = st.nsucc;
    r.nunsucc = st.nunsucc;
    r.pad = 0;
    results[w] = r;
    win_status[w] = 1;
    atomicAdd(done_count, 1);
  }
}

// ---------------------------------------------------------------------------------------
// k_eval (debug / parity): raw r and local Jacobian rows at the initial point.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(FOLD_THREADS) void k_eval(SolveParams prm, const TileDev* __restrict__ tiles,
                                                       uint32_t n_tiles, const WinDev* __restrict__ wins,
                                                       const PairDev* __restrict__ pairs,
                                                       const double* __restrict__ frames,
                                                       const double* __restrict__ init_x,
                                                       const double* __restrict__ m0, const double* __restrict__ m1,
                                                       double* __restrict__ res, double* __restrict__ jac) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * FOLD_WAVES + wv;
  if (t >= n_tiles) return;
  const TileDev tl = tiles[t];
  const WinDev win = wins[tl.win];
  const PairDev pd = pairs[tl.pair];
  const int M = win.M;
  const double* __restrict__ fr = frames + (size_t)MVO_FRAME_DOUBLES * win.frame_off;
  double zvec[MVO_MAXM][3], zt[MVO_MAXM], zs[MVO_MAXM], zsp[MVO_MAXM];
  for (int k = 0; k < M; ++k) {
    for (int j = 0; j < 3; ++j) zvec[k][j] = init_x[((size_t)tl.win * MVO_MAXM + k) * 4 + j];
    zt[k] = init_x[((size_t)tl.win * MVO_MAXM + k) * 4 + 3];
    zs[k] = d_scale(zt[k], prm.max_len, prm.min_len);
    zsp[k] = d_dscale(zt[k], prm.max_len, prm.min_len);
  }
  double u[3], nb;
  pair_unit_baseline(fr, M, (int)pd.a, (int)pd.b, zvec, zs, u, nb);
  const int a = pd.a, kmax = (int)pd.b < M - 1 ? (int)pd.b : M - 1;
  const double* __restrict__ R0 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f0;
  const double* __restrict__ R1 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f1;
  const int ncol = 3 * M;
  for (uint32_t i = lane; i < tl.count; i += 64) {
    const size_t o = (size_t)tl.src_start + i;
    const double* p0 = m0 + 3 * o;
    const double* p1 = m1 + 3 * o;
    const double px = R0[0] * p0[0] + R0[1] * p0[1] + R0[2] * p0[2];
    const double py = R0[3] * p0[0] + R0[4] * p0[1] + R0[5] * p0[2];
    const double pz = R0[6] * p0[0] + R0[7] * p0[1] + R0[8] * p0[2];
    const double qx = R1[0] * p1[0] + R1[1] * p1[1] + R1[2] * p1[2];
    const double qy = R1[3] * p1[0] + R1[4] * p1[1] + R1[5] * p1[2];
    const double qz = R1[6] * p1[0] + R1[7] * p1[1] + R1[8] * p1[2];
    const double cx = py * qz - pz * qy, cy = pz * qx - px * qz, cz = px * qy - py * qx;
    const double r = u[0] * cx + u[1] * cy + u[2] * cz;
    if (res) res[o] = r;
    if (jac) {
      double* __restrict__ row = jac + o * ncol;
      for (int c = 0; c < ncol; ++c) row[c] = 0.0;
      const double gx = (cx - u[0] * r) / nb, gy = (cy - u[1] * r) / nb, gz = (cz - u[2] * r) / nb;
      for (int k = a; k <= kmax; ++k) {
        const double vx = zvec[k][0], vy = zvec[k][1], vz = zvec[k][2];
        row[3 * k + 0] = zs[k] * (gx * (-2.0 * vy) + gy * (2.0 * vx));
        row[3 * k + 1] = zs[k] * (gx * (2.0 * vz) + gz * (-2.0 * vx));
        row[3 * k + 2] = zsp[k] * (vx * gx + vy * gy + vz * gz);
      }
    }
  }
}

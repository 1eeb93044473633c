// mvo_lm.h -- k_lm: the device-resident Levenberg-Marquardt loop (one workgroup per window).
//
// Restates ceres::Solve as the reference configures it (IR.cpp:63-76,231-233): TRUST_REGION +
// LEVENBERG_MARQUARDT, use_nonmonotonic_steps, jacobi scaling, Ceres 1.14 defaults for the rest
// (TrustRegionMinimizer / LevenbergMarquardtStrategy / TrustRegionStepEvaluator rules).
//
// Work split inside the workgroup:
//   all threads : per-pair geometry + projected moments  (thread per frame pair)   -> LDS
//                 class sums in pair order, 3x3 block products G_k^T W_kl G_l       -> LDS
//   wavefront 0 : the <= 15 x 15 damped normal equations, one matrix ROW PER LANE held in
//                 registers: Gauss-Jordan elimination with v_readlane broadcasts (no LDS, no
//                 barriers), accept/reject logic executed redundantly by all 64 lanes on
//                 wave-uniform scalars, per-frame retraction on lanes 0..M-1.
// The state lives in wave 0's registers for the whole loop; it touches HBM only when the
// launch begins/ends (pass mode re-enters once per observation pass).
#pragma once

#include "mvo_device.h"

// ---- wave-level helpers -----------------------------------------------------------------
__device__ __forceinline__ double bcast_lane(double v, int src_lane /* compile-time uniform */) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// compact column index of (frame k, component j); fix_length drops the t column
__device__ __forceinline__ int ccol(int k, int j, int fix) { return fix ? 2 * k + j : 3 * k + j; }
// full (3M) column of compact column i
__device__ __forceinline__ int fcol(int i, int fix) { return fix ? 3 * (i >> 1) + (i & 1) : i; }

// per-pair index of (a, b) in the reference's loop order
__device__ __forceinline__ int pair_index(int N, int a, int b) { return a * (N - 1) - (a * (a - 1)) / 2 + (b - a); }

struct LmShared {
  double zvec[MVO_MAXM][3], zt[MVO_MAXM], zs[MVO_MAXM], zsp[MVO_MAXM];
  double G[MVO_MAXM][9];                        // G[k][3*row + col]: columns (s_k jl0, s_k jl1, s'_k v_k)
  double cls[MVO_MAXM * (MVO_MAXM + 1)][10];    // class (a, j = min(b, M)): T(6) q(3) cost(1)
  double Hz[MVO_MAXC * MVO_MAXC], gz[MVO_MAXC]; // full 3M system at z
  double cost_z;
  int32_t go, phase;
};

// Register-resident solver state of wavefront 0.
struct LmRegs {
  // lane l < n: row l of J^T J (active columns), gradient, jacobi scale, LM diagonal
  double hrow[MVO_MAXC], g, js, dg, hdiag;
  // lane k < M: accepted / candidate / best point of refined frame k
  double xv[3], xt, cv[3], ct, bv[3], bt;
  // wave-uniform scalars
  double x_cost, x_norm, min_cost, mcc, radius, decf, gmn, initial_cost, final_cost;
  double se_min, se_cur, se_ref, se_cand, se_acc_ref, se_acc_cand;
  int32_t se_nonmono, reuse_diag, iteration, num_invalid, nsucc, nunsucc, status, termination, reason, phase,
      ntrace;
};

__device__ __forceinline__ void regs_load(LmRegs& r, const LmState* __restrict__ g, int lane) {
  const int l = lane < MVO_MAXC ? lane : 0, k = lane < MVO_MAXM ? lane : 0;
#pragma unroll
  for (int j = 0; j < MVO_MAXC; ++j) r.hrow[j] = g->H[l * MVO_MAXC + j];
  r.g = g->g[l];
  r.js = g->jscale[l];
  r.dg = g->diag[l];
  r.hdiag = g->H[l * MVO_MAXC + l];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r.xv[j] = g->x_vec[k][j];
    r.cv[j] = g->c_vec[k][j];
    r.bv[j] = g->b_vec[k][j];
  }
  r.xt = g->x_t[k];
  r.ct = g->c_t[k];
  r.bt = g->b_t[k];
  r.x_cost = g->x_cost; r.x_norm = g->x_norm; r.min_cost = g->min_cost; r.mcc = g->model_cost_change;
  r.radius = g->radius; r.decf = g->decrease_factor; r.gmn = g->grad_max_norm;
  r.initial_cost = g->initial_cost; r.final_cost = g->final_cost;
  r.se_min = g->se_min; r.se_cur = g->se_cur; r.se_ref = g->se_ref; r.se_cand = g->se_cand;
  r.se_acc_ref = g->se_acc_ref; r.se_acc_cand = g->se_acc_cand;
  r.se_nonmono = g->se_nonmono; r.reuse_diag = g->reuse_diagonal; r.iteration = g->iteration;
  r.num_invalid = g->num_invalid; r.nsucc = g->nsucc; r.nunsucc = g->nunsucc; r.status = g->status;
  r.termination = g->termination; r.reason = g->reason; r.phase = g->phase; r.ntrace = g->ntrace;
}

__device__ __forceinline__ void regs_store(const LmRegs& r, LmState* __restrict__ g, int lane) {
  if (lane < MVO_MAXC) {
#pragma unroll
    for (int j = 0; j < MVO_MAXC; ++j) g->H[lane * MVO_MAXC + j] = r.hrow[j];
    g->g[lane] = r.g;
    g->jscale[lane] = r.js;
    g->diag[lane] = r.dg;
  }
  if (lane < MVO_MAXM) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      g->x_vec[lane][j] = r.xv[j];
      g->c_vec[lane][j] = r.cv[j];
      g->b_vec[lane][j] = r.bv[j];
    }
    g->x_t[lane] = r.xt;
    g->c_t[lane] = r.ct;
    g->b_t[lane] = r.bt;
  }
  if (lane == 0) {
    g->x_cost = r.x_cost; g->x_norm = r.x_norm; g->min_cost = r.min_cost; g->model_cost_change = r.mcc;
    g->radius = r.radius; g->decrease_factor = r.decf; g->grad_max_norm = r.gmn;
    g->initial_cost = r.initial_cost; g->final_cost = r.final_cost;
    g->se_min = r.se_min; g->se_cur = r.se_cur; g->se_ref = r.se_ref; g->se_cand = r.se_cand;
    g->se_acc_ref = r.se_acc_ref; g->se_acc_cand = r.se_acc_cand;
    g->se_nonmono = r.se_nonmono; g->reuse_diagonal = r.reuse_diag; g->iteration = r.iteration;
    g->num_invalid = r.num_invalid; g->nsucc = r.nsucc; g->nunsucc = r.nunsucc; g->status = r.status;
    g->termination = r.termination; g->reason = r.reason; g->phase = r.phase; g->ntrace = r.ntrace;
  }
}

// || a - b || over the program's parameters (vec always, t unless fix_length); lanes k < M
// hold frame k.  inf_norm: max |.|, else Euclidean.
__device__ __forceinline__ double frames_diff_norm(int lane, int M, int fix, const double av[3], double at,
                                                   const double bv[3], double bt, bool inf_norm) {
  double acc = 0.0;
  if (lane < M) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double d = fabs(av[j] - bv[j]);
      acc = inf_norm ? fmax(acc, d) : acc + d * d;
    }
    if (!fix) {
      const double d = fabs(at - bt);
      acc = inf_norm ? fmax(acc, d) : acc + d * d;
    }
  }
  return inf_norm ? wave_max(acc) : sqrt(wave_sum(acc));
}

// x (+) sign*delta for frame `lane`: delta lives one component per lane (compact columns)
__device__ __forceinline__ void frames_plus(int lane, int M, int fix, const double xv[3], double xt,
                                            double delta_lane, double sign, double ov[3], double& ot) {
  const int k = lane < M ? lane : 0;
  const double da = __shfl(delta_lane, ccol(k, 0, fix), 64);
  const double db = __shfl(delta_lane, ccol(k, 1, fix), 64);
  const double dt = __shfl(delta_lane, fix ? 0 : ccol(k, 2, fix), 64);
  d_plus(xv, sign * da, sign * db, ov);
  ot = fix ? xt : xt + sign * dt;
}

__device__ __forceinline__ void lm_terminate(LmRegs& r, int termination, int reason) {
  r.status = 1;
  r.termination = termination;
  r.reason = reason;
}

// Wave-0 control logic: Ceres 1.14 TrustRegionMinimizer::Minimize restated around the evaluation
// that was just completed at z (= x0 in PH_EVAL_X0, the candidate in PH_EVAL_CAND); the system
// assembled at z is in sh.Hz / sh.gz / sh.cost_z.  Every lane executes it; scalars are uniform.
// Returns 1 when a new candidate awaits evaluation (published in sh.zvec/zt), 0 when done.
__device__ int lm_advance(LmRegs& r, LmShared& sh, const SolveParams& prm, int M, int n, int lane,
                          TraceRec* __restrict__ trace) {
  const int fix = prm.fix_length;
  TraceRec it;
  it.cost = it.cost_change = it.gradient_max_norm = it.step_norm = it.relative_decrease = 0.0;
  it.trust_region_radius = 0.0;
  it.step_is_valid = it.step_is_successful = 0;

  bool take_z = false;
  const bool first = (r.phase == PH_EVAL_X0);
  const double cost_z = sh.cost_z;
  if (first) {
    // IterationZero
    take_z = true;
    r.iteration = 0;
    r.min_cost = DBL_MAX;
    r.radius = 1e4;  // initial_trust_region_radius
    r.decf = 2.0;
    r.reuse_diag = 0;
    r.num_invalid = 0;
    r.nsucc = r.nunsucc = 0;
    r.ntrace = 0;
    r.mcc = 0.0;
    r.cv[0] = r.xv[0]; r.cv[1] = r.xv[1]; r.cv[2] = r.xv[2]; r.ct = r.xt;  // z == x0
  } else {
    // ParameterToleranceReached
    it.step_norm = frames_diff_norm(lane, M, fix, r.xv, r.xt, r.cv, r.ct, false);
    if (it.step_norm <= prm.ptol * (r.x_norm + prm.ptol)) {
      lm_terminate(r, 0, 3);
      return 0;
    }
    // FunctionToleranceReached
    it.cost_change = r.x_cost - cost_z;
    if (fabs(it.cost_change) <= prm.ftol * r.x_cost) {
      lm_terminate(r, 0, 4);
      return 0;
    }
    // TrustRegionStepEvaluator::StepQuality
    const double rel = (r.se_cur - cost_z) / r.mcc;
    const double hist = (r.se_ref - cost_z) / (r.se_acc_ref + r.mcc);
    it.relative_decrease = fmax(rel, hist);
    it.step_is_valid = 1;
    if (it.relative_decrease > 1e-3) {  // min_relative_decrease
      take_z = true;
      it.step_is_successful = 1;
      // LevenbergMarquardtStrategy::StepAccepted
      const double q = 2.0 * it.relative_decrease - 1.0;
      r.radius = r.radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
      r.radius = fmin(1e16, r.radius);
      r.decf = 2.0;
      r.reuse_diag = 0;
      // TrustRegionStepEvaluator::StepAccepted
      r.se_cur = cost_z;
      r.se_acc_cand += r.mcc;
      r.se_acc_ref += r.mcc;
      if (r.se_cur < r.se_min) {
        r.se_min = r.se_cur;
        r.se_nonmono = 0;
        r.se_cand = r.se_cur;
        r.se_acc_cand = 0.0;
      } else {
        ++r.se_nonmono;
        if (r.se_cur > r.se_cand) {
          r.se_cand = r.se_cur;
          r.se_acc_cand = 0.0;
        }
      }
      if (r.se_nonmono == 5) {  // max_consecutive_nonmonotonic_steps
        r.se_ref = r.se_cand;
        r.se_acc_ref = r.se_acc_cand;
      }
    } else {
      // HandleUnsuccessfulStep + StepRejected
      r.radius = r.radius / r.decf;
      r.decf *= 2.0;
      r.reuse_diag = 1;
      it.cost = cost_z;
      it.gradient_max_norm = r.gmn;
    }
  }

  if (take_z) {
    // x <- z; EvaluateGradientAndJacobian = the system assembled at z
    r.xv[0] = r.cv[0]; r.xv[1] = r.cv[1]; r.xv[2] = r.cv[2]; r.xt = r.ct;
    r.x_cost = cost_z;
    {
      const double zero3[3] = {0.0, 0.0, 0.0};
      r.x_norm = frames_diff_norm(lane, M, fix, r.xv, r.xt, zero3, 0.0, false);
    }
    const int l = lane < n ? lane : 0;
    const int fl = fcol(l, fix);
#pragma unroll
    for (int j = 0; j < MVO_MAXC; ++j) r.hrow[j] = (lane < n && j < n) ? sh.Hz[fl * MVO_MAXC + fcol(j, fix)] : 0.0;
    r.g = lane < n ? sh.gz[fl] : 0.0;
    r.hdiag = lane < n ? sh.Hz[fl * MVO_MAXC + fl] : 0.0;
    if (first) {
      r.js = 1.0 / (1.0 + sqrt(r.hdiag));  // jacobi_scaling, fixed at iteration 0
      r.initial_cost = r.final_cost = r.x_cost;
      r.se_min = r.se_cur = r.se_ref = r.se_cand = r.x_cost;
      r.se_acc_ref = r.se_acc_cand = 0.0;
      r.se_nonmono = 0;
      it.step_is_valid = 1;
      it.step_is_successful = 1;
    }
    // gradient_max_norm = || x - Plus(x, -g) ||_inf
    double pv[3], pt;
    frames_plus(lane, M, fix, r.xv, r.xt, r.g, -1.0, pv, pt);
    r.gmn = frames_diff_norm(lane, M, fix, r.xv, r.xt, pv, pt, true);
    it.cost = r.x_cost;
    it.gradient_max_norm = r.gmn;
  }

  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
    if (it.step_is_successful) {
      ++r.nsucc;
      if (r.x_cost < r.min_cost) {
        r.min_cost = r.x_cost;
        r.bv[0] = r.xv[0]; r.bv[1] = r.xv[1]; r.bv[2] = r.xv[2]; r.bt = r.xt;
      }
    } else {
      ++r.nunsucc;
    }
    it.trust_region_radius = r.radius;
    if (trace && lane == 0 && r.ntrace < prm.trace_cap) trace[r.ntrace] = it;
    ++r.ntrace;
    if (it.cost < r.final_cost) r.final_cost = it.cost;

    if (r.iteration >= prm.max_it) {
      lm_terminate(r, 1, 1);
      return 0;
    }
    if (it.step_is_successful && it.gradient_max_norm <= prm.gtol) {
      lm_terminate(r, 0, 2);
      return 0;
    }
    if (r.radius <= 1e-32) {  // min_trust_region_radius
      lm_terminate(r, 0, 5);
      return 0;
    }

    // ---- next iteration: ComputeTrustRegionStep ----
    ++r.iteration;
    it.cost = it.cost_change = it.gradient_max_norm = it.step_norm = it.relative_decrease = 0.0;
    it.step_is_valid = it.step_is_successful = 0;

    if (!r.reuse_diag) r.dg = fmin(fmax(r.js * r.js * r.hdiag, 1e-6), 1e32);  // min/max_lm_diagonal
    r.reuse_diag = 1;
    // (Js^T Js + D^2) y = Js^T r with Js = J diag(jscale), D^2 = diag / radius; step = -y.
    // Ceres solves the equivalent least-squares problem by dense QR; here the reduced system is
    // eliminated in registers, one row per lane (Gauss-Jordan, no pivoting: the matrix is SPD).
    double as[MVO_MAXC], a[MVO_MAXC];
#pragma unroll
    for (int j = 0; j < MVO_MAXC; ++j) {
      const double jsj = bcast_lane(r.js, j);
      as[j] = (lane < n) ? r.js * r.hrow[j] * jsj : 0.0;  // lanes >= n carry no row
      a[j] = as[j];
      if (j == lane) a[j] = (lane < n) ? as[j] + r.dg / r.radius : 1.0;  // pad rows: identity
    }
    const double rhs0 = (lane < n) ? r.js * r.g : 0.0;
    double rhs = rhs0, mydiag = 1.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < MVO_MAXC; ++j) {
      const double piv = bcast_lane(a[j], j);
      if (!(piv > 0.0) || isinf(piv)) ok = false;
      const double f = (lane == j) ? 0.0 : a[j] / piv;
      if (lane == j) mydiag = piv;
#pragma unroll
      for (int k = j + 1; k < MVO_MAXC; ++k) a[k] -= f * bcast_lane(a[k], j);
      rhs -= f * bcast_lane(rhs, j);
    }
    double step = -(rhs / mydiag);
    if (lane >= n) step = 0.0;
    if (__any(!isfinite(step))) ok = false;
    double mcc = 0.0;
    if (ok) {
      // model_cost_change = -(Js step)^T (r + Js step/2) = -step^T Js^T r - step^T Js^T Js step / 2
      double row = 0.0;
#pragma unroll
      for (int j = 0; j < MVO_MAXC; ++j) row += as[j] * bcast_lane(step, j);
      const double lin = wave_sum(step * rhs0);
      const double quad = wave_sum(step * row);
      mcc = -(lin + 0.5 * quad);
    }
    r.mcc = mcc;
    const bool valid = ok && mcc > 0.0;
    if (!valid) {
      // ---- HandleInvalidStep ----
      if (++r.num_invalid >= 5) {  // max_num_consecutive_invalid_steps
        lm_terminate(r, 2, 6);
        return 0;
      }
      r.radius *= 0.5;  // StepIsInvalid
      r.reuse_diag = 1;
      it.cost = r.x_cost;
      it.gradient_max_norm = r.gmn;
      continue;
    }
    r.num_invalid = 0;
    // ---- ComputeCandidatePoint: delta = step o jscale; candidate = Plus(x, delta) ----
    frames_plus(lane, M, fix, r.xv, r.xt, step * r.js, 1.0, r.cv, r.ct);
    if (lane < M) {
      sh.zvec[lane][0] = r.cv[0];
      sh.zvec[lane][1] = r.cv[1];
      sh.zvec[lane][2] = r.cv[2];
      sh.zt[lane] = r.ct;
    }
    r.phase = PH_EVAL_CAND;
    return 1;
  }
}

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
  const int lane = tid & 63;
  const bool wave0 = tid < 64;
  LmState* __restrict__ gst = states + w;
  if (gst->status != 0) return;  // uniform: every thread reads the same word
  const WinDev win = wins[w];
  const int M = win.M, N = win.N, P = win.P;
  const int fix = prm.fix_length;
  const int n = fix ? 2 * M : 3 * M;
  const double* __restrict__ fr = frames + (size_t)MVO_FRAME_DOUBLES * win.frame_off;
  const PairDev* __restrict__ wp = pairs + win.pair_off;

  LmRegs r;
  if (wave0) {
    regs_load(r, gst, lane);
    if (r.phase == PH_START) {
      // IR.cpp:95-109: vec_k from the frame, t_k = reverseScale(scale_k) (host libm, uploaded)
      const int k = lane < M ? lane : 0;
      r.xv[0] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 0];
      r.xv[1] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 1];
      r.xv[2] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 2];
      r.xt = init_x[((size_t)w * MVO_MAXM + k) * 4 + 3];
    }
    const bool at_x = (r.phase != PH_EVAL_CAND);
    if (lane < M) {
      sh.zvec[lane][0] = at_x ? r.xv[0] : r.cv[0];
      sh.zvec[lane][1] = at_x ? r.xv[1] : r.cv[1];
      sh.zvec[lane][2] = at_x ? r.xv[2] : r.cv[2];
      sh.zt[lane] = at_x ? r.xt : r.ct;
    }
    if (lane == 0) sh.phase = r.phase;
  }
  __syncthreads();

  bool need_pass = false;
  if (sh.phase == PH_START) {
    if (wave0) r.phase = PH_EVAL_X0;
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
            for (int rr = 0; rr < 3; ++rr)
              for (int c = 0; c < 3; ++c)
                WG[3 * rr + c] = Wm[3 * rr] * sh.G[l][c] + Wm[3 * rr + 1] * sh.G[l][3 + c] + Wm[3 * rr + 2] * sh.G[l][6 + c];
            for (int i = 0; i < 3; ++i)
              for (int jj = 0; jj < 3; ++jj)
                blk[3 * i + jj] = sh.G[k][i] * WG[jj] + sh.G[k][3 + i] * WG[3 + jj] + sh.G[k][6 + i] * WG[6 + jj];
          }
          if (k == l) {
            const double* Ts = sh.cls[k * MC + k];
            const double Wm[9] = {Ts[0], Ts[1], Ts[2], Ts[1], Ts[3], Ts[4], Ts[2], Ts[4], Ts[5]};
            double WG[9];
            for (int rr = 0; rr < 3; ++rr)
              for (int c = 0; c < 3; ++c)
                WG[3 * rr + c] = Wm[3 * rr] * sh.G[k][c] + Wm[3 * rr + 1] * sh.G[k][3 + c] + Wm[3 * rr + 2] * sh.G[k][6 + c];
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
    // ---------------- wavefront 0: accept/reject, damping, next candidate ----------------
    if (wave0) {
      const int go = lm_advance(r, sh, prm, M, n, lane, (w == 0) ? trace : nullptr);
      if (lane == 0) sh.go = go;
    }
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
  if (wave0) {
    regs_store(r, gst, lane);
    if (r.status != 0) {
      LmResult* __restrict__ res = results + w;
      if (lane < MVO_MAXM) {
        res->vec[lane][0] = r.bv[0];
        res->vec[lane][1] = r.bv[1];
        res->vec[lane][2] = r.bv[2];
        res->t[lane] = r.bt;
      }
      if (lane == 0) {
        res->initial_cost = r.initial_cost;
        res->final_cost = r.final_cost;
        res->termination = r.termination;
        res->reason = r.reason;
        res->iterations = r.ntrace - 1;
        res->nsucc = r.nsucc;
        res->nunsucc = r.nunsucc;
        res->pad = 0;
        win_status[w] = 1;
        atomicAdd(done_count, 1);
      }
    }
  }
}

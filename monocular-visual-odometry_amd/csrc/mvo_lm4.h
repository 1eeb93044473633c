// mvo_lm4.h -- k_lm4: the device-resident Levenberg-Marquardt loop for BATCHES of small windows
// (at most 64 frame pairs each, no loss, sums folded before the launch): FOUR windows per wavefront.
//
// Why: in k_lm<64> the LM step (decision, damped 15 x 15 solve, candidate; lm_decide / lm_step) is half
// of an evaluation's instruction stream and uses 15 of the 64 lanes.  A wavefront issues one instruction
// per ~4.4 cycles whatever the lane count, so four windows side by side -- window q in the 16-lane DPP
// row q -- take the step for the price of one.  The evaluation itself (per-pair moments, class sums,
// 3 x 3 blocks: already 35..60 lanes wide) runs two windows at a time, side by side in one instruction
// stream (lm4_evaluate2): the stages of k_lm<64>, same operations in the same order.
//
// The step is the same restatement of Ceres 1.14's TrustRegionMinimizer as lm_decide / lm_step
// (mvo_lm.h), written without branches: what is wave-uniform there is row-uniform here, every `if` is a
// select under the row's condition.  Same arithmetic, operation for operation: a window solved here and
// a window solved by k_lm<64> take the same decisions and agree to rounding (tests/test_gpu_lm4.py).
//
// One launch runs its windows to termination (no re-entry, no state in HBM).  Replaces ceres::Solve,
// IterativeRefinement.cpp:231-233, for mvo_hip_refine_batch.
#pragma once

#include "mvo_lm.h"

constexpr int LM4_WINDOWS = 4;

struct alignas(16) Lm4Win {  // one window, kept across evaluations
  double zvec[MVO_MAXM][3], zt[MVO_MAXM], zs[MVO_MAXM], zsp[MVO_MAXM];
  double G[MVO_MAXM][9];
  double H[2][MVO_MAXC * MVO_MAXC];  // the accepted system and the one assembled at z (roles swap: hsel)
  double gz[MVO_MAXC];
  double cost;
};

// scratch of ONE evaluation.  Two of them: the windows are evaluated two at a time, side by side in one instruction
// stream (lm4_evaluate2), so that one window's LDS round trips and dependent fp64 chains run under the other's.
struct alignas(16) Lm4Eval {     // (rows of ten doubles: every row starts on a 16-byte boundary)
  double kk[MVO_MAXM][10];       // pair (k, k)
  double Cd[MVO_MAXM][10];       // C1[k][k + 1] of lm_prefix_suffix: what the diagonal blocks add to D[k][k + 1]
  double D[MVO_MAXM * MCX][10];  // D[k][j] = sum_{a <= k} sum_{j' >= j} class (a, j')
  double pairbuf[64 * 10];
};
constexpr int LM4_SIDE = 2;

struct alignas(16) Lm4Shared {
  Lm4Win win[LM4_WINDOWS];
  Lm4Eval ev[LM4_SIDE];
};

// totals over each 16-lane row, left in every lane of the row
__device__ __forceinline__ double row_sum_all(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ double row_max_all(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return v;
}
__device__ __forceinline__ bool row_any(bool b) { return row_max_all(b ? 1.0 : 0.0) > 0.0; }

__device__ __forceinline__ double sel(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ int sel(bool c, int a, int b) { return c ? a : b; }

// frames_plus / frames_diff_* of mvo_lm.h per row: l = lane & 15 is the frame / the compact column
__device__ __forceinline__ void frames_plus4(int lane, int M, int fix, const double xv[3], double xt,
                                             double delta_lane, double sign, double ov[3], double& ot) {
  const int l = lane & 15, base = lane & 48;
  const int k = l < M ? l : 0;
  const double da = __shfl(delta_lane, base + ccol(k, 0, fix), 64);
  const double db = __shfl(delta_lane, base + ccol(k, 1, fix), 64);
  const double dt = __shfl(delta_lane, base + (fix ? 0 : ccol(k, 2, fix)), 64);
  d_plus(xv, sign * da, sign * db, ov);
  ot = fix ? xt : xt + sign * dt;
}
__device__ __forceinline__ double frames_diff_sq4(int lane, int M, int fix, const double av[3], double at,
                                                  const double bv[3], double bt) {
  const double d0 = av[0] - bv[0], d1 = av[1] - bv[1], d2 = av[2] - bv[2], d3 = fix ? 0.0 : at - bt;
  const double acc = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
  return row_sum_all((lane & 15) < M ? acc : 0.0);
}
__device__ __forceinline__ double frames_diff_max4(int lane, int M, int fix, const double av[3], double at,
                                                   const double bv[3], double bt) {
  const double d3 = fix ? 0.0 : fabs(at - bt);
  const double acc = fmax(fmax(fabs(av[0] - bv[0]), fabs(av[1] - bv[1])), fmax(fabs(av[2] - bv[2]), d3));
  return row_max_all((lane & 15) < M ? acc : 0.0);
}

// gj_eliminate of mvo_lm.h with the pivot kept per row (row_bcast instead of v_readlane)
template <int J>
__device__ __forceinline__ void gj_eliminate4(double (&a)[MVO_MAXC], double& rhs, double& myrinv, unsigned& phimax,
                                              int l, double ddl, double piv, double rinv) {
  if constexpr (J < MVO_MAXC) {
    const unsigned hi = (unsigned)__double2hiint(piv);
    phimax = hi > phimax ? hi : phimax;
    int l_here = l;
    asm volatile("" : "+v"(l_here));
    const bool me = l_here == J;
    const double f = me ? 0.0 : a[J] * rinv;
    myrinv = me ? rinv : myrinv;
    double piv_next = 1.0, rinv_next = 1.0;
    if constexpr (J + 1 < MVO_MAXC) {
      a[J + 1] -= f * row_bcast<J>(a[J + 1]);
      piv_next = row_bcast<J + 1>(a[J + 1] + ddl);
      rinv_next = fast_rcp(piv_next);
      gj_row_update<J, J + 2>(a, f);
    }
    rhs -= f * row_bcast<J>(rhs);
    gj_eliminate4<J + 1>(a, rhs, myrinv, phimax, l, ddl, piv_next, rinv_next);
  }
}

// Solver state of the lane's row (window): as LmRegs, every scalar row-uniform.
struct Lm4Regs {
  double g, js, dg, hdiag;
  double xv[3], xt, cv[3], ct, bv[3], bt;
  double x_cost, x_norm, min_cost, mcc, radius, decf, gmn, initial_cost, final_cost;
  double se_min, se_cur, se_ref, se_cand, se_acc_ref, se_acc_cand;
  int32_t se_nonmono, reuse_diag, iteration, num_invalid, nsucc, nunsucc, status, termination, reason, phase, ntrace,
      hsel;
};

__device__ __forceinline__ void lm4_terminate(Lm4Regs& r, bool c, int termination, int reason) {
  r.status = sel(c, 1, r.status);
  r.termination = sel(c, termination, r.termination);
  r.reason = sel(c, reason, r.reason);
}

// One LM step for the four rows.  act: the row's window was just evaluated at its published point.
// Returns (per lane, row-uniform) whether the row published a new candidate.
__device__ __forceinline__ bool lm4_advance(Lm4Regs& r, Lm4Shared& sh, const SolveParams& prm, int M, int n, int lane,
                                            bool act) {
  const int fix = prm.fix_length;
  const int row = lane >> 4, l = lane & 15;
  Lm4Win& win = sh.win[row];
  // ---------------- lm_decide ----------------
  double it_cost = 0.0, it_gmn = 0.0;
  bool it_succ = false;
  const bool first = r.phase == PH_EVAL_X0;
  const double cost_raw = win.cost;
  const double cost_z = (first || isfinite(cost_raw)) ? cost_raw : DBL_MAX;
  const bool bad0 = act && first && !isfinite(cost_z);
  const bool init0 = act && first && !bad0;
  const bool cand = act && !first;
  {
    // the initial point failed to evaluate
    r.iteration = sel(bad0, 0, r.iteration);
    r.ntrace = sel(bad0, 1, r.ntrace);
    r.nsucc = sel(bad0 || init0, 0, r.nsucc);
    r.nunsucc = sel(bad0 || init0, 0, r.nunsucc);
    r.initial_cost = sel(bad0, cost_z, r.initial_cost);
    r.final_cost = sel(bad0, cost_z, r.final_cost);
#pragma unroll
    for (int j = 0; j < 3; ++j) r.bv[j] = sel(bad0, r.xv[j], r.bv[j]);
    r.bt = sel(bad0, r.xt, r.bt);
    lm4_terminate(r, bad0, 2, 8);
    // IterationZero
    r.iteration = sel(init0, 0, r.iteration);
    r.min_cost = sel(init0, DBL_MAX, r.min_cost);
    r.radius = sel(init0, 1e4, r.radius);
    r.decf = sel(init0, 2.0, r.decf);
    r.reuse_diag = sel(init0, 0, r.reuse_diag);
    r.num_invalid = sel(init0, 0, r.num_invalid);
    r.ntrace = sel(init0, 0, r.ntrace);
    r.mcc = sel(init0, 0.0, r.mcc);
#pragma unroll
    for (int j = 0; j < 3; ++j) r.cv[j] = sel(init0, r.xv[j], r.cv[j]);
    r.ct = sel(init0, r.xt, r.ct);
  }
  // ParameterToleranceReached / FunctionToleranceReached
  const double sn2 = frames_diff_sq4(lane, M, fix, r.xv, r.xt, r.cv, r.ct);
  const double step_norm = prm.ptol > 0.0 ? sqrt(sn2) : 0.0;
  const bool ptol_hit = cand && (prm.ptol > 0.0 ? step_norm <= prm.ptol * (r.x_norm + prm.ptol) : sn2 <= 0.0);
  const double cost_change = r.x_cost - cost_z;
  const bool ftol_hit = cand && !ptol_hit && fabs(cost_change) <= prm.ftol * r.x_cost;
  lm4_terminate(r, ptol_hit, 0, 3);
  lm4_terminate(r, ftol_hit, 0, 4);
  const bool judged = cand && !ptol_hit && !ftol_hit;
  // TrustRegionStepEvaluator::StepQuality
  const double rel = (r.se_cur - cost_z) * fast_rcp(r.mcc);
  const double hist = (r.se_ref - cost_z) * fast_rcp(r.se_acc_ref + r.mcc);
  const double it_rel = fmax(rel, hist);
  const bool accept = judged && it_rel > 1e-3;  // min_relative_decrease
  const bool reject = judged && !accept;
  {
    // LevenbergMarquardtStrategy::StepAccepted
    const double q = 2.0 * it_rel - 1.0;
    const double rad_acc = fmin(1e16, r.radius * fast_rcp(fmax(1.0 / 3.0, 1.0 - q * q * q)));
    const double rad_rej = r.radius * fast_rcp(r.decf);  // a power of two: exact
    r.radius = sel(accept, rad_acc, sel(reject, rad_rej, r.radius));
    r.decf = sel(accept, 2.0, sel(reject, r.decf * 2.0, r.decf));
    r.reuse_diag = sel(accept, 0, sel(reject, 1, r.reuse_diag));
    // TrustRegionStepEvaluator::StepAccepted
    const double se_cur = cost_z;
    double se_acc_cand = r.se_acc_cand + r.mcc, se_acc_ref = r.se_acc_ref + r.mcc;
    double se_min = r.se_min, se_cand = r.se_cand, se_ref = r.se_ref;
    int se_nonmono = r.se_nonmono;
    const bool lower = se_cur < se_min;
    const bool above = !lower && se_cur > se_cand;
    se_min = sel(lower, se_cur, se_min);
    se_nonmono = sel(lower, 0, se_nonmono + 1);
    se_cand = sel(lower || above, se_cur, se_cand);
    se_acc_cand = sel(lower || above, 0.0, se_acc_cand);
    const bool reset_ref = se_nonmono == 5;  // max_consecutive_nonmonotonic_steps
    se_ref = sel(reset_ref, se_cand, se_ref);
    se_acc_ref = sel(reset_ref, se_acc_cand, se_acc_ref);
    r.se_cur = sel(accept, se_cur, r.se_cur);
    r.se_acc_cand = sel(accept, se_acc_cand, r.se_acc_cand);
    r.se_acc_ref = sel(accept, se_acc_ref, r.se_acc_ref);
    r.se_min = sel(accept, se_min, r.se_min);
    r.se_cand = sel(accept, se_cand, r.se_cand);
    r.se_ref = sel(accept, se_ref, r.se_ref);
    r.se_nonmono = sel(accept, se_nonmono, r.se_nonmono);
    // HandleUnsuccessfulStep + StepRejected
    it_cost = sel(reject, cost_z, it_cost);
    it_gmn = sel(reject, r.gmn, it_gmn);
  }
  const bool take_z = init0 || accept;
  it_succ = take_z;
  {
    // x <- z
#pragma unroll
    for (int j = 0; j < 3; ++j) r.xv[j] = sel(take_z, r.cv[j], r.xv[j]);
    r.xt = sel(take_z, r.ct, r.xt);
    r.x_cost = sel(take_z, cost_z, r.x_cost);
    if (prm.ptol > 0.0) {  // ||x|| is only ever read by the parameter-tolerance test
      const double zero3[3] = {0.0, 0.0, 0.0};
      const double xn = sqrt(frames_diff_sq4(lane, M, fix, r.xv, r.xt, zero3, 0.0));
      r.x_norm = sel(take_z, xn, r.x_norm);
    }
    r.initial_cost = sel(init0, r.x_cost, r.initial_cost);
    r.final_cost = sel(init0, r.x_cost, r.final_cost);
    r.se_min = sel(init0, r.x_cost, r.se_min);
    r.se_cur = sel(init0, r.x_cost, r.se_cur);
    r.se_ref = sel(init0, r.x_cost, r.se_ref);
    r.se_cand = sel(init0, r.x_cost, r.se_cand);
    r.se_acc_ref = sel(init0, 0.0, r.se_acc_ref);
    r.se_acc_cand = sel(init0, 0.0, r.se_acc_cand);
    r.se_nonmono = sel(init0, 0, r.se_nonmono);
    it_cost = sel(take_z, r.x_cost, it_cost);
  }
  // ---------------- lm_step ----------------
  {
    // EvaluateGradientAndJacobian = the system assembled at z (the buffer hsel does not select)
    const double* __restrict__ Hnew = win.H[r.hsel ^ 1];
    const int fl = fcol(l < n ? l : 0, fix);
    const double gz = win.gz[fl], hd = Hnew[fl * MVO_MAXC + fl];
    r.hsel = sel(take_z, r.hsel ^ 1, r.hsel);
    r.g = sel(take_z, l < n ? gz : 0.0, r.g);
    r.hdiag = sel(take_z, l < n ? hd : 0.0, r.hdiag);
    if (__any(init0)) {  // (a square root and a division: only in the turn that follows a row's first evaluation)
      const double js0 = 1.0 / (1.0 + sqrt(fmax(r.hdiag, 0.0)));  // jacobi_scaling, fixed at iteration 0
      r.js = sel(init0, js0, r.js);
    }
    // gradient_max_norm = || x - Plus(x, -g) ||_inf
    double pv[3], pt;
    frames_plus4(lane, M, fix, r.xv, r.xt, r.g, -1.0, pv, pt);
    const double gmn = frames_diff_max4(lane, M, fix, r.xv, r.xt, pv, pt);
    r.gmn = sel(take_z, gmn, r.gmn);
    it_gmn = sel(take_z, r.gmn, it_gmn);
  }
  bool run = act && !bad0 && !ptol_hit && !ftol_hit;  // rows that go on to the end of the iteration
  bool go = false;
  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
    {
      const bool succ = run && it_succ, unsucc = run && !it_succ;
      r.nsucc += succ ? 1 : 0;
      r.nunsucc += unsucc ? 1 : 0;
      const bool better = succ && r.x_cost < r.min_cost;
      r.min_cost = sel(better, r.x_cost, r.min_cost);
#pragma unroll
      for (int j = 0; j < 3; ++j) r.bv[j] = sel(better, r.xv[j], r.bv[j]);
      r.bt = sel(better, r.xt, r.bt);
      r.ntrace += run ? 1 : 0;
      r.final_cost = sel(run, fmin(r.final_cost, it_cost), r.final_cost);
    }
    const bool stop_it = run && r.iteration >= prm.max_it;
    const bool stop_g = run && !stop_it && it_succ && it_gmn <= prm.gtol;
    const bool stop_r = run && !stop_it && !stop_g && r.radius <= 1e-32;  // min_trust_region_radius
    lm4_terminate(r, stop_it, 1, 1);
    lm4_terminate(r, stop_g, 0, 2);
    lm4_terminate(r, stop_r, 0, 5);
    const bool stp = run && !stop_it && !stop_g && !stop_r;
    if (!__any(stp)) break;
    // ---- next iteration: ComputeTrustRegionStep ----
    r.iteration += stp ? 1 : 0;
    it_cost = sel(stp, 0.0, it_cost);
    it_gmn = sel(stp, 0.0, it_gmn);
    it_succ = it_succ && !stp;
    {
      const double dg = fmin(fmax(r.js * r.js * r.hdiag, 1e-6), 1e32);  // min/max_lm_diagonal
      r.dg = sel(stp && !r.reuse_diag, dg, r.dg);
      r.reuse_diag = sel(stp, 1, r.reuse_diag);
    }
    const double jsm = l < n ? r.js : 0.0;
    double a[MVO_MAXC];
    {
      const double* __restrict__ Hacc = win.H[r.hsel];
      const int fl = fcol(l < n ? l : 0, fix);
      double hrow[MVO_MAXC];
#pragma unroll
      for (int j = 0; j < MVO_MAXC; ++j) hrow[j] = Hacc[fl * MVO_MAXC + (fix ? 3 * (j >> 1) + (j & 1) : j)];
      gj_setup<0>(a, hrow, jsm);
    }
    const double dd = r.dg * fast_rcp(r.radius);
    const double ddl = l < n ? dd : 1.0;
    const double rhs0 = jsm * r.g;
    double rhs = rhs0, myrinv = 1.0;
    unsigned phimax = 0;
    {
      const double piv0 = row_bcast<0>(a[0] + ddl);
      gj_eliminate4<0>(a, rhs, myrinv, phimax, l, ddl, piv0, fast_rcp(piv0));
    }
    const double step = l < n ? -(rhs * myrinv) : 0.0;
    const bool ok = phimax < 0x7ff00000u && !row_any(!isfinite(step));
    const double mcc = ok ? 0.5 * row_sum_all(step * (dd * step - rhs0)) : 0.0;
    r.mcc = sel(stp, mcc, r.mcc);
    const bool valid = stp && ok && mcc > 0.0;
    const bool invalid = stp && !valid;
    // ---- HandleInvalidStep ----
    r.num_invalid = sel(invalid, r.num_invalid + 1, sel(valid, 0, r.num_invalid));
    const bool give_up = invalid && r.num_invalid >= 5;  // max_num_consecutive_invalid_steps
    lm4_terminate(r, give_up, 2, 6);
    const bool retry = invalid && !give_up;
    r.radius = sel(retry, r.radius * 0.5, r.radius);  // StepIsInvalid
    it_cost = sel(retry, r.x_cost, it_cost);
    it_gmn = sel(retry, r.gmn, it_gmn);
    // ---- ComputeCandidatePoint: delta = step o jscale; candidate = Plus(x, delta) ----
    {
      double cv[3], ct;
      frames_plus4(lane, M, fix, r.xv, r.xt, step * r.js, 1.0, cv, ct);
#pragma unroll
      for (int j = 0; j < 3; ++j) r.cv[j] = sel(valid, cv[j], r.cv[j]);
      r.ct = sel(valid, ct, r.ct);
      if (valid) lm_publish_point(win, prm, M, l, r.cv, r.ct);
      r.phase = sel(valid, (int)PH_EVAL_CAND, r.phase);
      go = go || valid;
    }
    run = retry;
    if (!__any(run)) break;
  }
  return go;
}

#ifdef MVO_LM_STAMPS
// diagnostic build: cycles of workgroup 0 per phase, slots 24..31 of mvo_lm_stamps (24 = turns of the main loop,
// 25 per-pair stage, 26 tables, 27 unused, 28 blocks + gradient, 29 lm4_advance, 30 prologue, 31 epilogue); kept in
// registers and written once at the end (a global read-modify-write per stamp would put an L2 round trip into every phase)
struct Lm4Stamps {
  unsigned long long prev, acc[8];
};
#define LM4_STAMP(i)                                                \
  do {                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    lm4_st_.acc[(i)] += now_ - lm4_st_.prev;                        \
    lm4_st_.prev = now_;                                            \
  } while (0)
#define LM4_STAMP_PARAM , Lm4Stamps& lm4_st_
#define LM4_STAMP_ARG , lm4_st_
#else
#define LM4_STAMP(i) do { } while (0)
#define LM4_STAMP_PARAM
#define LM4_STAMP_ARG
#endif

// what a lane does in the stages of an evaluation (fixed for the launch)
struct Lm4Role {
  int kT, vT;             // tables: frame k, value v (lanes < 10 M)
  bool onT;
  double mT[MVO_MAXM];    // 1.0 for the frames a <= k that exist, else 0.0
  BlockRole br;           // blocks / gradient
};
__device__ __forceinline__ Lm4Role lm4_role(int M, int lane) {
  Lm4Role ro;
  ro.kT = lane / 10;
  ro.vT = lane - 10 * ro.kT;
  ro.onT = ro.kT < M;  // (M <= 5: lanes >= 50 are off)
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) ro.mT[a] = (a <= ro.kT && a < M) ? 1.0 : 0.0;
  ro.br = block_role(M, lane);
  return ro;
}

// One evaluation of the windows QA and QA + 1 at their published points: the stages of k_lm<64> -- per-pair projected
// moments; class sums, their prefix over a and suffix over j; 3 x 3 blocks and gradient -- for both windows in ONE
// instruction stream.  Every stage is written loads first, for both windows, then arithmetic, then stores under one
// predicate, so that the scheduler sees two independent chains in each basic block.
//
// Tables (stages 2 and 3 of lm_prefix_suffix in one): lane (k, v) adds up, for every class j, the classes (a, j) of the
// frames a <= min(k, j) in the order a = 0, 1, ... (= C1[k][j]), then the suffix over j from M down (= D[k][j]); the
// long class j = M is the sum of the pairs b = M .. N-2 in pair order.  Same operations in the same order as
// lm_prefix_suffix<true>, one LDS round trip instead of N - M + 2.  Frames a > k take part through a factor 0.0 in a
// fused multiply-add (exact: C + x * 1 = C + x, C + x * 0 = C), frames a >= M and classes j >= M that do not exist read
// pair 0 instead (a valid row; their factor is 0.0 / their sum is never used).
template <int QA>
__device__ __forceinline__ void lm4_evaluate2(Lm4Shared& sh, const PairIn& inA, const PairIn& inB, bool my_ok0, int lane,
                                              int M, int N, int hselA, int hselB, const Lm4Role& ro LM4_STAMP_PARAM) {
  constexpr int NW = LM4_SIDE;
  // ---- (1) per-pair projected moments ----
  {
    double o[NW][10];
    const int kmax = inA.b < M - 1 ? inA.b : M - 1;  // (a, b) of a lane's pair are the same in every window
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      double beta[3];
      free_baseline(sh.win[QA + w], inA.a, kmax, beta);
      pair_eval(w == 0 ? inA : inB, beta, 0, o[w]);
    }
    if (my_ok0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        double* __restrict__ dst = sh.ev[w].pairbuf + lane * 10;
#pragma unroll
        for (int e = 0; e < 10; ++e) dst[e] = o[w][e];
      }
      if (inA.a == inA.b) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
          for (int e = 0; e < 10; ++e) sh.ev[w].kk[inA.a][e] = o[w][e];
      }
    }
  }
  wave0_sync();
  LM4_STAMP(1);
  // ---- (2) + (3) class sums, prefix over a, suffix over j ----
  {
    const int cnt = N - 1 - M;
    int ra[MVO_MAXM];  // first pair of frame a (wave-uniform)
#pragma unroll
    for (int a = 0; a < MVO_MAXM; ++a) ra[a] = a < M ? pair_index(N, a, a) : 0;
    double x[NW][MVO_MAXM][MVO_MAXM], L[NW][MVO_MAXM];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const double* __restrict__ pb = sh.ev[w].pairbuf + ro.vT;
#pragma unroll
      for (int a = 0; a < MVO_MAXM; ++a) {
#pragma unroll
        for (int j = a; j < MVO_MAXM; ++j) x[w][a][j] = pb[(j < M ? ra[a] + (j - a) : 0) * 10];
        L[w][a] = 0.0;
      }
    }
#pragma unroll 4
    for (int i = 0; i < cnt; ++i) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int a = 0; a < MVO_MAXM; ++a)
          L[w][a] += sh.ev[w].pairbuf[((a < M ? ra[a] + (M - a) : 0) + i) * 10 + ro.vT];
    }
    double C[NW][MVO_MAXM], acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
#pragma unroll
      for (int j = 0; j < MVO_MAXM; ++j) {
        double c = 0.0;
#pragma unroll
        for (int a = 0; a <= j; ++a) c = fma(x[w][a][j], ro.mT[a], c);
        C[w][j] = c;
      }
      double cl = 0.0;
#pragma unroll
      for (int a = 0; a < MVO_MAXM; ++a) cl = fma(L[w][a], ro.mT[a], cl);
      acc[w] = 0.0 + cl;
    }
    if (ro.onT) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        Lm4Eval& ev = sh.ev[w];
        ev.D[ro.kT * MCX + M][ro.vT] = acc[w];
        double cd = 0.0;  // C1[k][k + 1]
#pragma unroll
        for (int j = MVO_MAXM - 1; j >= 0; --j) {
          if (j < M) {
            acc[w] += C[w][j];
            ev.D[ro.kT * MCX + j][ro.vT] = acc[w];
          }
          cd = (j == ro.kT + 1) ? C[w][j] : cd;
        }
        ev.Cd[ro.kT][ro.vT] = cd;
        if (ro.kT == M - 1 && ro.vT == 9) sh.win[QA + w].cost = acc[w];  // all classes: D[M-1][0], cost entry
      }
    }
  }
  wave0_sync();
  LM4_STAMP(2);
  // ---- (4) 3 x 3 blocks of J^T J (lm_block_rows<0>) and the gradient (lm_gradient_entry) ----
  {
    const int k4 = ro.br.k4, l4 = ro.br.l4, i4 = ro.br.i4;
    const bool diag = k4 == l4;
    const int hs[NW] = {hselA, hselB};
    // a block-row lane reads the values 0..5 of its rows (the moments), a gradient lane 6..8 (P S u): six loads from a
    // lane-dependent offset (a gradient lane's last three run into the next row -- inside Lm4Eval -- and are ignored)
    const int vo = lane < ro.br.nB4 ? 0 : 6;
    double Wv[NW][6], Ts[NW][6], gk[NW][3], Gl[NW][9];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const Lm4Eval& ev = sh.ev[w];
      const Lm4Win& win = sh.win[QA + w];
      const double* __restrict__ wsrc = ev.D[k4 * MCX + (diag ? k4 + 1 : l4)] + vo;
      const double* __restrict__ csrc = ev.Cd[k4 > 0 ? k4 - 1 : 0] + vo;
      const double* __restrict__ ksrc = ev.kk[k4] + vo;
#pragma unroll
      for (int v = 0; v < 6; ++v) {
        const double cv = csrc[v];  // (loaded by every lane: a load under the condition would be a branch)
        Wv[w][v] = wsrc[v] + ((diag && k4 > 0) ? cv : 0.0);
        Ts[w][v] = ksrc[v];
      }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) gk[w][rr] = win.G[k4][3 * rr + i4];
#pragma unroll
      for (int e = 0; e < 9; ++e) Gl[w][e] = win.G[l4][e];
    }
    if (lane < ro.br.nB4) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        double val[3];
        {
          double gw[3];  // (G_k^T W)[i][m]
#pragma unroll
          for (int m = 0; m < 3; ++m)
            gw[m] = gk[w][0] * Wv[w][sym6(0, m)] + gk[w][1] * Wv[w][sym6(1, m)] + gk[w][2] * Wv[w][sym6(2, m)];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) val[jj] = gw[0] * Gl[w][jj] + gw[1] * Gl[w][3 + jj] + gw[2] * Gl[w][6 + jj];
        }
        {  // the pair (k, k): direction columns only (rows i < 2 of the diagonal blocks)
          double gw[3];
#pragma unroll
          for (int m = 0; m < 3; ++m)
            gw[m] = gk[w][0] * Ts[w][sym6(0, m)] + gk[w][1] * Ts[w][sym6(1, m)] + gk[w][2] * Ts[w][sym6(2, m)];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const double add = val[jj] + (gw[0] * Gl[w][jj] + gw[1] * Gl[w][3 + jj] + gw[2] * Gl[w][6 + jj]);
            val[jj] = (diag && i4 < 2) ? add : val[jj];
          }
        }
        double* __restrict__ Hw = sh.win[QA + w].H[hs[w] ^ 1];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
          if (!diag || jj >= i4) {  // diagonal blocks: the upper entry is mirrored, H stays symmetric
            Hw[(3 * k4 + i4) * MVO_MAXC + 3 * l4 + jj] = val[jj];
            Hw[(3 * l4 + jj) * MVO_MAXC + 3 * k4 + i4] = val[jj];
          }
      }
    } else if (lane < ro.br.nB4 + 3 * M) {
      // gradient entry (k, i) = G_k^T y_k, y_k = D[k][k+1] + C1[k-1][k] + (i < 2 ? pair (k, k) : 0), values 6..8
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        double gi = 0.0;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          // Wv = D[k][k+1] + (k > 0 ? C1[k-1][k] : 0) as lm_gradient_entry forms it (these lanes have k4 == l4)
          const double yv = Wv[w][v] + (i4 < 2 ? Ts[w][v] : 0.0);
          gi += gk[w][v] * yv;
        }
        sh.win[QA + w].gz[3 * k4 + i4] = gi;
      }
    }
  }
  wave0_sync();
  LM4_STAMP(4);
}

// Workgroup b solves the windows win_base + 4 b .. + 3 (those below win_end).
// MS: 0 = any shape; 5 = windows with M = 5 refined frames and a free length (the reference's usual window,
// BASELINE configs[2]): M, the 15 unknowns and every "does this frame / class exist" test are compile-time constants.
// host_marks: the result records lie in pinned host memory and the host may poll their marks (few windows).
// mark: what a finished window's record carries in `pad`.  done_count may be null: then the marks are the only
// completion signal (the host passes a new mark every solve and checks every record after the copy) -- the windows of
// a batch end together, and a thousand wavefronts' atomics on ONE address take 6 us at the end of the kernel (four
// thousand, one per window: 17 us).
// wpw: windows per wavefront, 4 or 2 (rows 2 and 3 then carry no window: one side-by-side evaluation per turn instead
// of two, under the same step -- batches that would leave most SIMDs without a wavefront at four per wavefront).
template <int MS>
__global__ __launch_bounds__(64) void k_lm4(SolveParams prm, int win_end, int unit_slots, int host_marks, int mark, int wpw,
                                            const double* __restrict__ frames, const double* __restrict__ init_x,
                                            const PairDev* __restrict__ pairs, const double* partial,
                                            LmResult* __restrict__ results, int32_t* __restrict__ win_status,
                                            int32_t* __restrict__ done_count) {
  __shared__ Lm4Shared sh;
#ifdef MVO_LM_STAMPS
  Lm4Stamps lm4_st_;
  lm4_st_.prev = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 8; ++i) lm4_st_.acc[i] = 0;
#endif
  const int lane = threadIdx.x, row = lane >> 4, l = lane & 15;
  if constexpr (MS != 0) {  // (prm is this thread's copy: what the host guarantees becomes a constant for every callee)
    prm.M = MS;
    prm.fix_length = 0;
  }
  const int M = prm.M, N = prm.N, P = prm.P;
  const int fix = prm.fix_length;
  const int n = fix ? 2 * M : 3 * M;
  const int wbase = prm.win_base + wpw * (int)blockIdx.x;
  const int w_row = wbase + row;
  const bool row_exists = row < wpw && w_row < win_end;
  const int w_ld = row_exists ? w_row : win_end - 1;  // loads of a row without a window: any valid one

  // every global load of the prologue first: the starting points and the pairs' inputs of the four windows
  double x0w[4];
  {
    const int k = l < M ? l : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) x0w[e] = init_x[((size_t)w_ld * MVO_MAXM + k) * 4 + e];
  }
  const int my_p0 = lane < P ? lane : 0;
  const bool my_ok0 = lane < P;
  PairIn in[LM4_WINDOWS];
  if (unit_slots) {
#pragma unroll
    for (int q = 0; q < LM4_WINDOWS; ++q) {
      const int wq = (q < wpw && wbase + q < win_end) ? wbase + q : win_end - 1;
      pair_load(in[q], my_p0, N, M, true, pairs + (size_t)wq * P, partial, (uint32_t)wq * (uint32_t)P,
                frames + (size_t)MVO_FRAME_DOUBLES * wq * N);
    }
  } else {
    // the fold's slots (no ticket tails in front of this kernel): the four windows' pair records in one round trip,
    // then their slots side by side, i-th slot of each in one more -- every pair adds its own up in slot order
    PairDev pd[LM4_WINDOWS];
    uint32_t most = 0;
#pragma unroll
    for (int q = 0; q < LM4_WINDOWS; ++q) {
      const int wq = (q < wpw && wbase + q < win_end) ? wbase + q : win_end - 1;
      pd[q] = pairs[(size_t)wq * P + my_p0];
    }
#pragma unroll
    for (int q = 0; q < LM4_WINDOWS; ++q) {
      in[q].a = (int)pd[q].a;
      in[q].b = (int)pd[q].b;
#pragma unroll
      for (int v = 0; v < 7; ++v) in[q].S[v] = 0.0;
      most = max(most, pd[q].slot_end - pd[q].slot_begin);
    }
    for (uint32_t i = 0; i < most; ++i) {
      double t[LM4_WINDOWS][7];
#pragma unroll
      for (int q = 0; q < LM4_WINDOWS; ++q) {
        const uint32_t sl = pd[q].slot_begin + i;
        const double* ps = partial + (size_t)(sl < pd[q].slot_end ? sl : 0u) * 8;  // (slot 0 exists: this pair or another has slots)
#pragma unroll
        for (int v = 0; v < 7; ++v) t[q][v] = ps[v];
      }
#pragma unroll
      for (int q = 0; q < LM4_WINDOWS; ++q)
        if (pd[q].slot_begin + i < pd[q].slot_end) {
#pragma unroll
          for (int v = 0; v < 7; ++v) in[q].S[v] += t[q][v];
        }
    }
#pragma unroll
    for (int q = 0; q < LM4_WINDOWS; ++q) {
      const int wq = (q < wpw && wbase + q < win_end) ? wbase + q : win_end - 1;
      in[q].off[0] = in[q].off[1] = in[q].off[2] = 0.0;
      if (in[q].b >= M) {  // IR.cpp:137-144: only frame b itself
        const double* __restrict__ f = frames + (size_t)MVO_FRAME_DOUBLES * wq * N + MVO_FRAME_DOUBLES * in[q].b;
        const double nb = f[12];
        in[q].off[0] = nb * f[9];
        in[q].off[1] = nb * f[10];
        in[q].off[2] = nb * f[11];
      }
    }
  }
  // both system buffers of every window start from zero (see k_lm: the elimination reads whole rows)
  for (int i = lane; i < LM4_WINDOWS * 2 * MVO_MAXC * MVO_MAXC; i += 64) {
    const int q = i / (2 * MVO_MAXC * MVO_MAXC), e = i - q * (2 * MVO_MAXC * MVO_MAXC);
    (&sh.win[q].H[0][0])[e] = 0.0;
  }
  if (lane < LM4_WINDOWS) sh.win[lane].cost = 0.0;
  // frames k >= M are never published; free_baseline multiplies them by s = 0, so they must hold numbers
  if (lane < LM4_WINDOWS * MVO_MAXM) {
    Lm4Win& wz = sh.win[lane / MVO_MAXM];
    const int k = lane % MVO_MAXM;
    wz.zs[k] = 0.0;
    wz.zvec[k][0] = wz.zvec[k][1] = wz.zvec[k][2] = 0.0;
  }
  wave0_sync();
  if (row_exists && l == 0) win_status[w_row] = 0;

  Lm4Regs r;
  r.g = r.js = r.dg = r.hdiag = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) r.xv[j] = r.cv[j] = r.bv[j] = 0.0;
  r.xt = r.ct = r.bt = 0.0;
  r.x_cost = r.x_norm = r.min_cost = r.mcc = r.radius = r.decf = r.gmn = r.initial_cost = r.final_cost = 0.0;
  r.se_min = r.se_cur = r.se_ref = r.se_cand = r.se_acc_ref = r.se_acc_cand = 0.0;
  r.se_nonmono = r.reuse_diag = r.iteration = r.num_invalid = r.nsucc = r.nunsucc = 0;
  r.termination = r.reason = r.ntrace = r.hsel = 0;
  r.status = row_exists ? 0 : 1;
  r.phase = PH_EVAL_X0;
  r.xv[0] = x0w[0]; r.xv[1] = x0w[1]; r.xv[2] = x0w[2]; r.xt = x0w[3];  // IR.cpp:95-109
  lm_publish_point(sh.win[row], prm, M, l, r.xv, r.xt);
  wave0_sync();
  const Lm4Role ro = lm4_role(M, lane);

  unsigned long long need = __ballot(r.status == 0 && l == 0);  // bit 16 q: window q awaits an evaluation
  LM4_STAMP(6);
  while (need) {
#ifdef MVO_LM_STAMPS
    lm4_st_.acc[0] += 1;
#endif
    // two windows at a time; a window of the two that has terminated is evaluated along (nobody reads the result)
    if (need & 0x00010001ull)
      lm4_evaluate2<0>(sh, in[0], in[1], my_ok0, lane, M, N, __builtin_amdgcn_readlane(r.hsel, 0),
                       __builtin_amdgcn_readlane(r.hsel, 16), ro LM4_STAMP_ARG);
    if (need & 0x0001000100000000ull)
      lm4_evaluate2<2>(sh, in[2], in[3], my_ok0, lane, M, N, __builtin_amdgcn_readlane(r.hsel, 32),
                       __builtin_amdgcn_readlane(r.hsel, 48), ro LM4_STAMP_ARG);
    const bool act = (need >> (16 * row)) & 1ull;
    const bool go = lm4_advance(r, sh, prm, M, n, lane, act);
    wave0_sync();
    need = __ballot(go && l == 0);
    LM4_STAMP(5);
  }

  LM4_STAMP(3);
  if (row_exists) {
    LmResult* __restrict__ res = results + w_row;
    if (l < MVO_MAXM) {
      // Ceres hands the lowest-cost iterate back only when the summary is usable; after FAILURE the
      // user's arrays keep their INITIAL values (see k_lm)
      const bool usable = r.termination != 2;
      const double bt = usable ? r.bt : x0w[3];
      res->vec[l][0] = usable ? r.bv[0] : x0w[0];
      res->vec[l][1] = usable ? r.bv[1] : x0w[1];
      res->vec[l][2] = usable ? r.bv[2] : x0w[2];
      res->scale[l] = d_scale(bt, prm.max_len, prm.min_len);
    }
    if (l == 0) {
      res->initial_cost = r.initial_cost;
      res->final_cost = r.final_cost;
      res->termination = r.termination;
      res->reason = r.reason;
      res->iterations = r.ntrace - 1;
      res->nsucc = r.nsucc;
      res->nunsucc = r.nunsucc;
    }
  }
  if (host_marks) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // see k_lm: the record is out before its mark
    if (row_exists && l == 0)
      __hip_atomic_store(&results[w_row].pad, mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if (row_exists && l == 0) {
    // records in device memory, copied by the stream behind this kernel: no system-scope release (an L2 write-back per
    // wavefront, a thousand of them arriving together)
    results[w_row].pad = mark;
  }
  if (row_exists && l == 0) win_status[w_row] = 1;
  if (done_count != nullptr && lane == 0)
    atomicAdd(done_count, win_end - wbase < wpw ? win_end - wbase : wpw);
  LM4_STAMP(7);
#ifdef MVO_LM_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mvo_lm_stamps[24 + i] += lm4_st_.acc[i];
  }
#endif
}

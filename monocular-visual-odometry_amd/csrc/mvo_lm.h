// mvo_lm.h -- k_lm: the device-resident Levenberg-Marquardt loop (one workgroup per window).
//
// Restates ceres::Solve as the reference configures it (IR.cpp:63-76,231-233): TRUST_REGION +
// LEVENBERG_MARQUARDT, use_nonmonotonic_steps, jacobi scaling, Ceres 1.14 defaults for the rest
// (TrustRegionMinimizer / LevenbergMarquardtStrategy / TrustRegionStepEvaluator rules).
//
// Two shapes of workgroup:
//   k_lm<64>   one wavefront, windows of <= 64 frame pairs (the batched case: thousands of windows):
//              lane per pair -> LDS, class sums by lanes (j, value)
//   k_lm<256>  four wavefronts with fixed roles (one per SIMD: 512 registers each), any window:
//              wavefronts 1..3 own the pairs (a, b >= M) -- the one long class of frame a; frames
//              a = w - 1 and a = w + 2 -- up to four per lane and frame, and reduce their ten sums per
//              frame with a register butterfly (v_permlane32/16_swap + DPP, no LDS); wavefront 0
//              owns the <= 15 pairs (a, b < M), each a class of its own, and the LM step.  The two
//              roles are two separate loops that meet at the workgroup barriers, so that neither
//              carries the other's registers (per-pair inputs of eight slots vs the solver state).
// After that, wavefront 0 alone: prefix / suffix sums of the class totals (W_kl, y_k), the 3 x 3
// block products G_k^T W_kl G_l, and the LM step: the <= 15 x 15 damped normal equations one matrix
// ROW PER LANE in registers, Gauss-Jordan with DPP row broadcasts and v_readlane pivots, Ceres'
// accept / reject logic on wave-uniform scalars, per-frame retraction on lanes 0..M-1.  The solver
// state lives in wavefront 0's registers from the first evaluation to the last; it touches LDS / HBM
// only when a launch begins or ends (pass mode re-enters once per observation pass).
#pragma once

#include "mvo_device.h"
#include "mvo_kernels.h"
#include "mvo_grid.h"

#include <stddef.h>

#define MVO_LM_NT_LARGE 256

// Diagnostic build only (-DMVO_LM_STAMPS): thread 0 of window 0 accumulates the shader-clock
// cycles spent in each phase of an LM iteration into a buffer of its own (mvo_lm_stamps).
#ifdef MVO_LM_STAMPS
__device__ unsigned long long mvo_lm_stamps[32];  // 0..15 k_lm, 16..23 the persistent solver, 24..31 k_lm4
// accumulated in LDS with fire-and-forget ds_add (a global read-modify-write per stamp would put
// an L2 round trip into every measured phase); flushed once when the kernel ends
#define MVO_STAMP_ADD(i, v) \
  __hip_atomic_fetch_add(&stamp_acc_[(i)], (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define MVO_STAMP(i)                                                        \
  do {                                                                      \
    if (w == 0 && tid == 0) {                                               \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
      if ((i) > 0) MVO_STAMP_ADD(i, now_ - stamp_prev_);                    \
      else MVO_STAMP_ADD(0, 1);                                             \
      stamp_prev_ = now_;                                                   \
    }                                                                       \
  } while (0)
// finer stamps inside lm_advance (slots 8..15); `stamp_on` = window 0
#define MVO_ASTAMP(i)                                                       \
  do {                                                                      \
    if (stamp_on && lane == 0) {                                            \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
      MVO_STAMP_ADD(i, now_ - astamp_prev_);                                \
      astamp_prev_ = now_;                                                  \
    }                                                                       \
  } while (0)
#define MVO_STAMP_PARAM , unsigned long long* stamp_acc_
#define MVO_STAMP_ARG , stamp_acc_
// the persistent robust solver's phases (workgroup 0, thread 0), slots 16..23 of mvo_lm_stamps
#define MVO_RSTAMP(i)                                                                        \
  do {                                                                                       \
    if (PERSIST && blockIdx.x == 0 && threadIdx.x == 0) {                                    \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                          \
      mvo_lm_stamps[16 + (i)] += now_ - rstamp_prev_;                                        \
      rstamp_prev_ = now_;                                                                   \
    }                                                                                        \
  } while (0)
#else
#define MVO_RSTAMP(i) do { } while (0)
#define MVO_STAMP(i) do { } while (0)
#define MVO_ASTAMP(i) do { } while (0)
#define MVO_STAMP_PARAM
#define MVO_STAMP_ARG
#endif

// ---- wave-level helpers -----------------------------------------------------------------
// Ordering point between two phases that only wavefront 0 executes: LDS operations of one
// wavefront complete in order, so no s_barrier is needed -- only the compiler must not move LDS
// accesses across it.
__device__ __forceinline__ void wave0_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double bcast_lane(double v, int src_lane /* compile-time uniform */) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// Broadcast of lane SRC (0..15) of each 16-lane row to all lanes of that row (row_newbcast, the
// one DPP control gfx950 offers for 64-bit operands), result stays in VGPRs -- used for the elimination's data flow (all matrix rows live
// in row 0).  Values that steer control flow use bcast_lane (v_readlane: truly wave-uniform).
template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + SRC, 0xf, 0xf, true);  // one v_mov_b64_dpp
}

// DPP permutations inside each 16-lane row (no LDS traffic): xor 1, xor 2, half-row mirror, row
// mirror.  After the four steps every lane of a row holds the row's total.  All state of k_lm's
// dense algebra lives in row 0 (lanes 0..14 = matrix rows, lanes 0..4 = refined frames).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row0_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return bcast_lane(v, 0);
}
__device__ __forceinline__ double row0_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return bcast_lane(v, 0);
}

// (swap_add32 / swap_add16: mvo_wave.h)
// 10 values x 64 lanes -> 10 totals.  After the call lane L holds the total of value
//   v(L) = ((L >> 5) & 1) * 4 + ((L >> 4) & 1) * 2 + ((L >> 3) & 1)        in `t8` (values 0..7) and of
//   value 8 + ((L >> 5) & 1)                                               in `t2` (values 8, 9),
// identical within the 8 lanes (t8) / 32 lanes (t2) of a group.  Fixed order => reproducible.
__device__ __forceinline__ void wave_reduce10(const double (&v)[10], int lane, double& t8, double& t2) {
  double w4[4], w2[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = swap_add32(v[j], v[j + 4]);      // lanes < 32: value j, lanes >= 32: value j + 4
  double u = swap_add32(v[8], v[9]);                                   // lanes < 32: value 8, lanes >= 32: value 9
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = swap_add16(w4[j], w4[j + 2]);    // even rows: value j (+4), odd rows: value j + 2 (+4)
  const bool b3 = lane & 8;
  {
    const double keep = b3 ? w2[1] : w2[0], send = b3 ? w2[0] : w2[1];
    t8 = keep + dpp_mov_f64<0x128>(send);  // row_ror:8 == lane xor 8
  }
  t8 += dpp_mov_f64<0xB1>(t8);
  t8 += dpp_mov_f64<0x4E>(t8);
  t8 += dpp_mov_f64<0x141>(t8);
  u = swap_add16(u, u);                    // every lane: own row + the partner row (xor 16)
  u += dpp_mov_f64<0xB1>(u);
  u += dpp_mov_f64<0x4E>(u);
  u += dpp_mov_f64<0x141>(u);
  u += dpp_mov_f64<0x140>(u);
  t2 = u;
}

// s(t) and s'(t) of IR.cpp:298-309 with one exp and one reciprocal
__device__ __forceinline__ void d_scale_both(double t, double max_len, double min_len, double& s, double& sp) {
  const double e = fast_exp(-1.0 * t);
  if (isinf(e)) {
    s = min_len;
    sp = 0.0;
  } else {
    const double rg = fast_rcp(1.0 + e);  // 1 <= 1 + e < inf: the fast reciprocal is within 1 ulp
    s = min_len + ((max_len - min_len) * rg);
    sp = ((max_len - min_len) * rg) * (rg * e);
  }
}

// compact column index of (frame k, component j); fix_length drops the t column
__device__ __forceinline__ int ccol(int k, int j, int fix) { return fix ? 2 * k + j : 3 * k + j; }
// full (3M) column of compact column i
__device__ __forceinline__ int fcol(int i, int fix) { return fix ? 3 * (i >> 1) + (i & 1) : i; }

// per-pair index of (a, b) in the reference's loop order
__device__ __forceinline__ int pair_index(int N, int a, int b) { return a * (N - 1) - (a * (a - 1)) / 2 + (b - a); }

constexpr int MCX = MVO_MAXM + 1;  // classes per frame a: j = min(b, M) in 0..M

struct LmShared {
  LmState st;                                   // image of the solver state between launches (HBM <-> LDS at begin / end)
  double zvec[MVO_MAXM][3], zt[MVO_MAXM], zs[MVO_MAXM];
  double G[MVO_MAXM][9];                        // G[k][3*row + col]: columns (s_k jl0, s_k jl1, s'_k v_k)
  double cls[MVO_MAXM * MCX][10];               // class (a, j = min(b, M)): T(6) q(3) cost(1)   (k_lm<256>)
  double kk[MVO_MAXM][10];                      // the pair (k, k) alone: direction columns only (see below)
  // prefix of the classes over a, C1[k][j] = sum_{a<=k} cls[a][j], and its suffix over j, D[k][j] = sum_{j'>=j}
  // C1[k][j'].  k_lm<256>: every helper wavefront forms its OWN copy right behind barrier B1 (the same 70
  // instructions, the same bits) and multiplies its share of the system out of it, while wavefront 0 decides
  // on the point -- no wavefront waits for another one's tables (one barrier and 1.1 k cycles per evaluation
  // less than with wavefront 0 forming them for everybody).  k_lm<64>: copy 0.
  double C1[3][MVO_MAXM * MCX][10];
  double D[3][MVO_MAXM * MCX][10];
  double Hz[MVO_MAXC * MVO_MAXC], gz[MVO_MAXC]; // second system buffer (st.H is the first) and the gradient at z
  double gmn_z;                                 // k_lm<256>: gradient max norm at z (helper wavefront)
  int32_t go, phase, hsel, pad_;
  // k_lm<256> with the step split off (LmSolver below): verdict -> solver before barrier B1c, step -> control before B2
  double ctl_radius;
  int32_t ctl_take, ctl_done, ctl_first, ctl_reuse;
  double mcc_term[MVO_MAXC + 1];                // per row: the model cost change is half their sum
  double zsp[MVO_MAXM];                         // s'(t_k) of the published point (the control wavefront forms G from it)
  int32_t step_ok, pad2_;
};

static_assert(offsetof(LmShared, kk) == offsetof(LmShared, cls) + sizeof(double) * MVO_MAXM * MCX * 10,
              "cls and kk are copied in as one block (SolveParams::cls_ready)");
static_assert((MVO_MAXM * MCX + MVO_MAXM) * 10 <= MVO_CLS_DOUBLES, "class totals per window");

// The solver state in wavefront 0's registers.
struct LmRegs {
  // lane l < n: gradient, jacobi scale, LM diagonal, H_ll of active column l (row l of H is in LDS)
  double g, js, dg, hdiag;
  // lane k < M: accepted / candidate / best point of refined frame k
  double xv[3], xt, cv[3], ct, bv[3], bt;
  // wave-uniform scalars
  double x_cost, x_norm, min_cost, mcc, radius, decf, gmn, initial_cost, final_cost;
  double se_min, se_cur, se_ref, se_cand, se_acc_ref, se_acc_cand;
  int32_t se_nonmono, reuse_diag, iteration, num_invalid, nsucc, nunsucc, status, termination, reason, phase,
      ntrace, pending_invalid;
};

__device__ __forceinline__ void regs_load(LmRegs& r, const LmState& g, int fix, int n, int lane) {
  const int l = lane < n ? lane : 0, k = lane < MVO_MAXM ? lane : 0;
  r.g = g.g[l];
  r.js = g.jscale[l];
  r.dg = g.diag[l];
  r.hdiag = g.H[fcol(l, fix) * (MVO_MAXC + 1)];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r.xv[j] = g.x_vec[k][j];
    r.cv[j] = g.c_vec[k][j];
    r.bv[j] = g.b_vec[k][j];
  }
  r.xt = g.x_t[k];
  r.ct = g.c_t[k];
  r.bt = g.b_t[k];
  r.x_cost = g.x_cost; r.x_norm = g.x_norm; r.min_cost = g.min_cost; r.mcc = g.model_cost_change;
  r.radius = g.radius; r.decf = g.decrease_factor; r.gmn = g.grad_max_norm;
  r.initial_cost = g.initial_cost; r.final_cost = g.final_cost;
  r.se_min = g.se_min; r.se_cur = g.se_cur; r.se_ref = g.se_ref; r.se_cand = g.se_cand;
  r.se_acc_ref = g.se_acc_ref; r.se_acc_cand = g.se_acc_cand;
  r.se_nonmono = g.se_nonmono; r.reuse_diag = g.reuse_diagonal; r.iteration = g.iteration;
  r.num_invalid = g.num_invalid; r.nsucc = g.nsucc; r.nunsucc = g.nunsucc; r.status = g.status;
  r.termination = g.termination; r.reason = g.reason; r.phase = g.phase; r.ntrace = g.ntrace;
  r.pending_invalid = g.pending_invalid;
}

__device__ __forceinline__ void regs_store(const LmRegs& r, LmState& g, int lane) {
  if (lane < MVO_MAXC) {
    g.g[lane] = r.g;
    g.jscale[lane] = r.js;
    g.diag[lane] = r.dg;
  }
  if (lane < MVO_MAXM) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      g.x_vec[lane][j] = r.xv[j];
      g.c_vec[lane][j] = r.cv[j];
      g.b_vec[lane][j] = r.bv[j];
    }
    g.x_t[lane] = r.xt;
    g.c_t[lane] = r.ct;
    g.b_t[lane] = r.bt;
  }
  if (lane == 0) {
    g.x_cost = r.x_cost; g.x_norm = r.x_norm; g.min_cost = r.min_cost; g.model_cost_change = r.mcc;
    g.radius = r.radius; g.decrease_factor = r.decf; g.grad_max_norm = r.gmn;
    g.initial_cost = r.initial_cost; g.final_cost = r.final_cost;
    g.se_min = r.se_min; g.se_cur = r.se_cur; g.se_ref = r.se_ref; g.se_cand = r.se_cand;
    g.se_acc_ref = r.se_acc_ref; g.se_acc_cand = r.se_acc_cand;
    g.se_nonmono = r.se_nonmono; g.reuse_diagonal = r.reuse_diag; g.iteration = r.iteration;
    g.num_invalid = r.num_invalid; g.nsucc = r.nsucc; g.nunsucc = r.nunsucc; g.status = r.status;
    g.termination = r.termination; g.reason = r.reason; g.phase = r.phase; g.ntrace = r.ntrace;
    g.pending_invalid = r.pending_invalid;
  }
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

// Lanes k < M publish the point to evaluate and everything the evaluation needs per frame:
// s_k, s'_k and G_k = [s_k J_loc(v_k) | s'_k v_k], J_loc(v) = [[-2vy, 2vz], [2vx, 0], [0, -2vx]].
// Two halves: what the per-pair stage reads (point and s_k) and what only the product stage reads (t_k, G_k).
// k_lm<256> writes the second half behind barrier B2, while the class wavefronts are already at the pairs.
template <class SH>
__device__ __forceinline__ void lm_publish_point_early(SH& sh, const SolveParams& prm, int M, int lane,
                                                       const double v[3], double t, double& s, double& sp) {
  d_scale_both(t, prm.max_len, prm.min_len, s, sp);
  if (lane < M) {
    const int k = lane;
    sh.zvec[k][0] = v[0];
    sh.zvec[k][1] = v[1];
    sh.zvec[k][2] = v[2];
    sh.zs[k] = s;
  }
}
template <class SH>
__device__ __forceinline__ void lm_publish_point_late(SH& sh, int M, int lane, const double v[3], double t, double s,
                                                      double sp) {
  if (lane < M) {
    const int k = lane;
    sh.zt[k] = t;
    sh.G[k][0] = s * (-2.0 * v[1]); sh.G[k][1] = s * (2.0 * v[2]);  sh.G[k][2] = sp * v[0];
    sh.G[k][3] = s * (2.0 * v[0]);  sh.G[k][4] = 0.0;               sh.G[k][5] = sp * v[1];
    sh.G[k][6] = 0.0;               sh.G[k][7] = s * (-2.0 * v[0]); sh.G[k][8] = sp * v[2];
  }
}
template <class SH>
__device__ __forceinline__ void lm_publish_point(SH& sh, const SolveParams& prm, int M, int lane,
                                                 const double v[3], double t) {
  double s, sp;
  lm_publish_point_early(sh, prm, M, lane, v, t, s, sp);
  lm_publish_point_late(sh, M, lane, v, t, s, sp);
}

// a[j] = js_l H_lj js_j for matrix row l = lane.  jsm = the jacobi scale on active lanes and 0 on the
// others: inactive rows and columns come out as zeros without a select.
template <int J>
__device__ __forceinline__ void gj_setup(double (&a)[MVO_MAXC], const double (&hrow)[MVO_MAXC], double jsm) {
  if constexpr (J < MVO_MAXC) {
    a[J] = (jsm * hrow[J]) * row_bcast<J>(jsm);
    gj_setup<J + 1>(a, hrow, jsm);
  }
}

template <int J, int K>
__device__ __forceinline__ void gj_row_update(double (&a)[MVO_MAXC], double f) {
  if constexpr (K < MVO_MAXC) {
    a[K] -= f * row_bcast<J>(a[K]);
    gj_row_update<J, K + 1>(a, f);
  }
}

// Gauss-Jordan elimination without pivoting (the matrix is SPD): after column J every other row
// has a zero in that column.  The LM damping D^2 = diag / radius is never written into the matrix: a
// row's diagonal entry is not touched by the steps before its own, so the damping is added when the
// entry becomes the pivot (ddl = the lane's damping on active rows, 1 on the all-zero padding rows).  Each lane
// keeps the reciprocal of its own pivot (myrinv), which is also what the validity test looks at afterwards: a negative
// pivot leaves a negative reciprocal (-0 for -inf), an infinite one +0, a NaN a NaN, a zero one an infinite step.
// Look-ahead order: step J updates column J+1 first and starts the reciprocal of the NEXT pivot
// (readlane -> v_rcp -> two Newton steps, a ~100-cycle dependent chain) before it touches the
// remaining columns, whose independent updates then fill that latency.
// n (wave-uniform): the active rows; the steps from n on would run on identity padding (the reference's
// shipped window refines two frames: 6 of 15) and are skipped by one scalar branch.
template <int J, bool FULL = false>
__device__ __forceinline__ void gj_eliminate(double (&a)[MVO_MAXC], double& rhs, double& myrinv,
                                             int lane, double ddl, double piv, double rinv, int n) {
  if constexpr (J < MVO_MAXC) {
    if (!FULL && J >= n) return;  // FULL: the caller checked n == MVO_MAXC (five refined frames, all columns free)
    // `lane == J` is recomputed at every step: left to itself the compiler hoists the fifteen lane
    // masks out of the solver loop, 30 scalar registers that it then spills and reloads one by one
    int lane_here = lane;
    asm volatile("" : "+v"(lane_here));
    const bool me = lane_here == J;
    const double f = me ? 0.0 : a[J] * rinv;
    myrinv = me ? rinv : myrinv;
    double piv_next = 1.0, rinv_next = 1.0;
    if constexpr (J + 1 < MVO_MAXC) {
      a[J + 1] -= f * row_bcast<J>(a[J + 1]);
      piv_next = bcast_lane(a[J + 1] + ddl, J + 1);
      rinv_next = fast_rcp(piv_next);
      gj_row_update<J, J + 2>(a, f);
    }
    rhs -= f * row_bcast<J>(rhs);
    gj_eliminate<J + 1, FULL>(a, rhs, myrinv, lane, ddl, piv_next, rinv_next, n);
  }
}

// sum of squares of (a - b) over the program's parameters (vec always, t unless fix_length) /
// largest |a - b|; lanes k < M hold frame k
__device__ __forceinline__ double frames_diff_sq(int lane, int M, int fix, const double av[3], double at,
                                                 const double bv[3], double bt) {
  const double d0 = av[0] - bv[0], d1 = av[1] - bv[1], d2 = av[2] - bv[2], d3 = fix ? 0.0 : at - bt;
  const double acc = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
  return row0_sum(lane < M ? acc : 0.0);
}
__device__ __forceinline__ double frames_diff_max(int lane, int M, int fix, const double av[3], double at,
                                                  const double bv[3], double bt) {
  const double d3 = fix ? 0.0 : fabs(at - bt);
  const double acc = fmax(fmax(fabs(av[0] - bv[0]), fabs(av[1] - bv[1])), fmax(fabs(av[2] - bv[2]), d3));
  return row0_max(lane < M ? acc : 0.0);
}

// what the product stage reads / writes (lm_block_rows, lm_gradient_entry): one copy of the tables
struct LmTables {
  double (*G)[9];
  double (*kk)[10];
  double (*C1)[10];
  double (*D)[10];
  double* gz;
};
__device__ __forceinline__ LmTables lm_tables(LmShared& sh, int copy) {
  LmTables t;
  t.G = sh.G; t.kk = sh.kk; t.C1 = sh.C1[copy]; t.D = sh.D[copy]; t.gz = sh.gz;
  return t;
}

// Stages (2) and (3) of an evaluation by ONE wavefront.
// (2) classes in pair order and their prefix over a:  C1[k][j] = sum_{a <= k} cls[a][j].  Class (a, j),
//     j = min(b, M): j < M is the single pair b = j, j = M collects the pairs b = M .. N-2 (those with a
//     fixed-frame offset).  Lane (j, v) walks a; classes that do not exist (a > j, a >= M, j > M) count as
//     zero, so every entry of C1 is defined.
// (3) suffix over j:  D[k][j] = sum_{j' >= j} C1[k][j'], lane (k, v).  W_kl = D[k][l] for l > k.
// FROM_PAIRS (k_lm<64> evaluating its own pairs): the classes are read from the per-pair rows in pairbuf.
template <bool FROM_PAIRS>
__device__ __forceinline__ void lm_prefix_suffix(const double (*cls)[10], const double* pairbuf, double (*C1)[10],
                                                 double (*D)[10], int M, int N, int lane) {
  const int jA = lane / 10, vA = lane - 10 * jA;
  const int kB = lane < MVO_MAXM * 10 ? jA : 0, vB = vA;
  if (lane < MCX * 10) {
    double x[MVO_MAXM];
    if (!FROM_PAIRS) {
#pragma unroll
      for (int a = 0; a < MVO_MAXM; ++a) x[a] = cls[a * MCX + jA][vA];  // reads issue back to back
    } else {
      // pairs of class (a, j), a <= j: one for j < M, the N - 1 - M pairs b >= M for j = M; the
      // five frames' loads of one step are independent (clamped rows for the classes that do not exist)
      const int cnt = (jA < M) ? 1 : (jA == M ? N - 1 - M : 0);
      int row[MVO_MAXM];
#pragma unroll
      for (int a = 0; a < MVO_MAXM; ++a) {
        x[a] = 0.0;
        row[a] = (a < M && a <= jA && jA <= M) ? pair_index(N, a, jA) : -1;
      }
      for (int i = 0; i < cnt; ++i) {
#pragma unroll
        for (int a = 0; a < MVO_MAXM; ++a) {
          const double t = pairbuf[(row[a] >= 0 ? row[a] + i : 0) * 10 + vA];
          x[a] += row[a] >= 0 ? t : 0.0;
        }
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < MVO_MAXM; ++a) {
      acc += (a < M && a <= jA && jA <= M) ? x[a] : 0.0;
      C1[a * MCX + jA][vA] = acc;
    }
  }
  wave0_sync();  // (this wavefront's own LDS writes, read back below)
  if (lane < MVO_MAXM * 10) {
    double x[MCX];
#pragma unroll
    for (int j = 0; j < MCX; ++j) x[j] = C1[kB * MCX + j][vB];
    double acc = 0.0;
#pragma unroll
    for (int j = MCX - 1; j >= 0; --j) {
      acc += x[j];
      D[kB * MCX + j][vB] = acc;
    }
  }
  wave0_sync();
}

// Stages (2) and (3) in one for k_lm<64> evaluating its own pairs (the form k_lm4 uses, mvo_lm4.h): lane (k, v) adds
// up, for every class j, the classes (a, j) of the frames a <= min(k, j) in the order a = 0, 1, ... (= C1[k][j]) straight
// from the per-pair rows, then the suffix over j from M down (= D[k][j]) -- the same operations in the same order as
// lm_prefix_suffix<true>, ONE LDS round trip instead of N - M + 2 (the long class's pairs cost one each there).
// Frames a > k take part through a factor 0.0 in a fused multiply-add (exact); frames a >= M and classes j >= M that
// do not exist read pair 0 instead (a valid row: their factor is 0.0 / their sum is never used).  Written: D[k][0..M]
// for every k < MVO_MAXM (rows k >= M repeat row M - 1: the cost is read from D[MVO_MAXM - 1][0]) and C1[k][k + 1],
// which is all the product stage and the verdict read.
__device__ __forceinline__ void lm_tables_from_pairs(const double* pairbuf, double (*C1)[10], double (*D)[10], int M,
                                                     int N, int lane) {
  const int kT = lane / 10, vT = lane - 10 * kT;
  const bool onT = kT < MVO_MAXM;
  double mT[MVO_MAXM];
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) mT[a] = (a <= kT && a < M) ? 1.0 : 0.0;
  const int cnt = N - 1 - M;
  int ra[MVO_MAXM];  // first pair of frame a (wave-uniform)
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) ra[a] = a < M ? pair_index(N, a, a) : 0;
  const double* __restrict__ pb = pairbuf + vT;
  double x[MVO_MAXM][MVO_MAXM], L[MVO_MAXM];
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) {
#pragma unroll
    for (int j = a; j < MVO_MAXM; ++j) x[a][j] = pb[(j < M ? ra[a] + (j - a) : 0) * 10];
    L[a] = 0.0;
  }
#pragma unroll 4
  for (int i = 0; i < cnt; ++i) {
#pragma unroll
    for (int a = 0; a < MVO_MAXM; ++a) L[a] += pb[((a < M ? ra[a] + (M - a) : 0) + i) * 10];
  }
  double C[MVO_MAXM];
#pragma unroll
  for (int j = 0; j < MVO_MAXM; ++j) {
    double c = 0.0;
#pragma unroll
    for (int a = 0; a <= j; ++a) c = fma(x[a][j], mT[a], c);
    C[j] = c;
  }
  double cl = 0.0;
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) cl = fma(L[a], mT[a], cl);
  double acc = 0.0 + cl;
  if (onT) {
    D[kT * MCX + M][vT] = acc;
    double cd = 0.0;  // C1[k][k + 1]
#pragma unroll
    for (int j = MVO_MAXM - 1; j >= 0; --j) {
      if (j < M) {
        acc += C[j];
        D[kT * MCX + j][vT] = acc;
      }
      cd = (j == kT + 1) ? C[j] : cd;
    }
    if (kT + 1 < MVO_MAXM) C1[kT * MCX + kT + 1][vT] = cd;
  }
  wave0_sync();
}

// k_lm<256>, wavefront 0: the cost at z = the total of all classes' cost entries, in the order the tables
// would give it (D[M-1][0][9]: per j the prefix over a, then the suffix from j = M down to 0) -- the verdict
// on z needs nothing else, so wavefront 0 does not wait for any table.
__device__ __forceinline__ double lm_cost_total(const double (*cls)[10], int M, int lane) {
  const int j = lane < MCX ? lane : 0;
  double x[MVO_MAXM];
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) x[a] = cls[a * MCX + j][9];
  double col = 0.0;
#pragma unroll
  for (int a = 0; a < MVO_MAXM; ++a) col += (a < M && a <= j && j <= M) ? x[a] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int jj = MCX - 1; jj >= 0; --jj) acc += bcast_lane(col, jj);
  return acc;
}

// The current ceres::IterationSummary and the verdict on the evaluation just completed, handed from
// lm_decide to lm_step.
struct LmDecision {
  double cost_z, it_cost, it_cost_change, it_gmn, it_step_norm, it_rel;
  int it_valid, it_succ;
  bool first, take_z, done;
};

__device__ __forceinline__ void lm_trace_row(TraceRec* __restrict__ trace, const SolveParams& prm, int lane, int row,
                                             const LmDecision& d, double radius) {
  if (trace != nullptr && lane == 0 && row < prm.trace_cap) {
    TraceRec t;
    t.cost = d.it_cost; t.cost_change = d.it_cost_change; t.gradient_max_norm = d.it_gmn; t.step_norm = d.it_step_norm;
    t.relative_decrease = d.it_rel; t.trust_region_radius = radius; t.step_is_valid = d.it_valid;
    t.step_is_successful = d.it_succ;
    trace[row] = t;
  }
}

// Wave-0 control logic: Ceres 1.14 TrustRegionMinimizer::Minimize restated around the evaluation
// that was just completed at z (= x0 in PH_EVAL_X0, the candidate in PH_EVAL_CAND).  Two halves:
//   lm_decide  needs the COST at z only: tolerances, step quality, accept / reject, radius, and the part of
//              FinalizeIterationAndCheckIfMinimizerCanContinue that needs no gradient.  In k_lm<256> it runs
//              while the helper wavefronts multiply out the blocks of the system at z.
//   lm_step    needs the system: x <- z, gradient norm, termination tests, the damped solve, the next
//              candidate.  Returns 1 when a point awaits evaluation (published in sh.zvec / zs).
// The system assembled at z is in the buffer `hsel` does NOT select (hsel ? sh.Hz : st.H holds the
// ACCEPTED system), gradient in sh.gz.  Every lane executes both; scalars are uniform.
// Written for a short, STRAIGHT instruction stream (one wavefront issues an instruction per ~4.4 cycles
// whatever the dependences, and every loop-carried value of a nested loop costs register copies at its
// latch): there is no inner loop.  An invalid step (solver failure, model cost change <= 0) is the rare case
// and takes the long way round: nothing new is published, the point in LDS is evaluated once more, and the
// next lm_decide runs HandleInvalidStep instead of a verdict (LmRegs::pending_invalid) -- the same sequence
// of records, radii and counters as Ceres' loop, one wasted evaluation per invalid step.
__device__ __forceinline__ void lm_finalize_early(LmRegs& r, const LmDecision& d) {
  // FinalizeIterationAndCheckIfMinimizerCanContinue, the part that needs no gradient
  if (d.it_succ) {
    ++r.nsucc;
    if (r.x_cost < r.min_cost) {
      r.min_cost = r.x_cost;
      r.bv[0] = r.xv[0]; r.bv[1] = r.xv[1]; r.bv[2] = r.xv[2]; r.bt = r.xt;
    }
  } else {
    ++r.nunsucc;
  }
  r.final_cost = fmin(r.final_cost, d.it_cost);
}

__device__ __forceinline__ void lm_decide(LmRegs& r, double cost_raw, const SolveParams& prm, int M, int lane,
                                          TraceRec* __restrict__ trace, LmDecision& d MVO_STAMP_PARAM) {
  const int fix = prm.fix_length;
#ifdef MVO_LM_STAMPS
  const bool stamp_on = stamp_acc_ != nullptr;
  unsigned long long astamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
  const bool tracing = trace != nullptr && prm.trace_cap > 0;
  d.it_cost = d.it_cost_change = d.it_gmn = d.it_step_norm = d.it_rel = 0.0;
  d.it_valid = d.it_succ = 0;
  d.first = (r.phase == PH_EVAL_X0);
  d.done = false;
  d.take_z = false;
  d.cost_z = 0.0;
  if (r.pending_invalid) {
    // ---- HandleInvalidStep: the evaluation that just completed repeated a point and is ignored ----
    r.pending_invalid = 0;
    if (++r.num_invalid >= 5) {  // max_num_consecutive_invalid_steps
      lm_terminate(r, 2, 6);
      d.done = true;
    } else {
      r.radius *= 0.5;  // StepIsInvalid
      r.reuse_diag = 1;
      d.it_cost = r.x_cost;
      d.it_gmn = r.gmn;
      lm_finalize_early(r, d);
    }
    MVO_ASTAMP(8);
    return;
  }
  // Ceres 1.14 rejects non-finite residuals (ResidualBlock::Evaluate / IsEvaluationValid): at the
  // initial point the minimizer fails before any step; a candidate that "failed to evaluate" gets
  // cost = max and is rejected.  A NaN/Inf residual makes the summed cost non-finite.
  // cost_raw = all classes: prefix over every a, suffix from j = 0 (D[M-1][0][9], or lm_cost_total)
  const double cost_z = (d.first || isfinite(cost_raw)) ? cost_raw : DBL_MAX;
  d.cost_z = cost_z;
  d.take_z = d.first;
  if (d.first) {
    if (!isfinite(cost_z)) {
      r.iteration = 0;
      r.ntrace = 1;  // iteration 0 is the one (failed) record: summary.iterations = 0
      d.it_cost = cost_z;
      lm_trace_row(trace, prm, lane, 0, d, 0.0);
      r.nsucc = r.nunsucc = 0;
      r.initial_cost = r.final_cost = cost_z;
      r.bv[0] = r.xv[0]; r.bv[1] = r.xv[1]; r.bv[2] = r.xv[2]; r.bt = r.xt;  // written back untouched
      lm_terminate(r, 2, 8);
      d.take_z = false;
      d.done = true;
    } else {
      // IterationZero
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
    }
  } else {
    // ParameterToleranceReached: ||x - z|| <= ptol (||x|| + ptol).  With ptol == 0 that is "the step
    // vanished", no square root needed (sqrt is > 0 for every positive argument).
    const double sn2 = frames_diff_sq(lane, M, fix, r.xv, r.xt, r.cv, r.ct);
    if (tracing || prm.ptol > 0.0) d.it_step_norm = sqrt(sn2);
    const bool ptol_hit = prm.ptol > 0.0 ? d.it_step_norm <= prm.ptol * (r.x_norm + prm.ptol) : sn2 <= 0.0;
    // FunctionToleranceReached
    d.it_cost_change = r.x_cost - cost_z;
    const bool ftol_hit = fabs(d.it_cost_change) <= prm.ftol * r.x_cost;
    if (ptol_hit || ftol_hit) {
      lm_terminate(r, 0, ptol_hit ? 3 : 4);
      d.done = true;
    } else {
      // TrustRegionStepEvaluator::StepQuality
      const double rel = (r.se_cur - cost_z) * fast_rcp(r.mcc);
      const double hist = (r.se_ref - cost_z) * fast_rcp(r.se_acc_ref + r.mcc);
      d.it_rel = fmax(rel, hist);
      d.it_valid = 1;
      d.take_z = d.it_rel > 1e-3;  // min_relative_decrease
      if (d.take_z) {
        d.it_succ = 1;
        // LevenbergMarquardtStrategy::StepAccepted
        const double q = 2.0 * d.it_rel - 1.0;
        r.radius = fmin(1e16, r.radius * fast_rcp(fmax(1.0 / 3.0, 1.0 - q * q * q)));
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
        r.radius = r.radius * fast_rcp(r.decf);  // a power of two: exact
        r.decf *= 2.0;
        r.reuse_diag = 1;
        d.it_cost = cost_z;
        d.it_gmn = r.gmn;
      }
    }
  }
  if (d.take_z) {
    // x <- z (the system at z is taken over in lm_step)
    r.xv[0] = r.cv[0]; r.xv[1] = r.cv[1]; r.xv[2] = r.cv[2]; r.xt = r.ct;
    r.x_cost = cost_z;
    if (prm.ptol > 0.0) {  // ||x|| is only ever read by the parameter-tolerance test
      const double zero3[3] = {0.0, 0.0, 0.0};
      r.x_norm = sqrt(frames_diff_sq(lane, M, fix, r.xv, r.xt, zero3, 0.0));
    }
    if (d.first) {
      r.initial_cost = r.final_cost = r.x_cost;
      r.se_min = r.se_cur = r.se_ref = r.se_cand = r.x_cost;
      r.se_acc_ref = r.se_acc_cand = 0.0;
      r.se_nonmono = 0;
      d.it_valid = 1;
      d.it_succ = 1;
    }
    d.it_cost = r.x_cost;
  }
  if (!d.done) lm_finalize_early(r, d);
  MVO_ASTAMP(8);
}

// GMN_READY: the gradient max norm at z was left in sh.gmn_z by a helper wavefront (k_lm<256>).
// mcc_term (per lane): the model cost change is 1/2 of its sum over row 0; the sum, its sign test and with it the
// last word on the step's validity are left to lm_model_cost_change, which k_lm<256> runs behind barrier B2
// while the class wavefronts already evaluate the candidate.
template <bool GMN_READY>
__device__ __forceinline__ int lm_step(LmRegs& r, LmShared& sh, const SolveParams& prm, int M, int n, int lane,
                                       TraceRec* __restrict__ trace, int& hsel, LmDecision& d, double& mcc_term,
                                       double& pub_s, double& pub_sp MVO_STAMP_PARAM) {
  // returns 0: the window has terminated; 1: a new candidate is published (its second half -- pub_s, pub_sp --
  // is the caller's to write, lm_publish_point_late); 2: the step is invalid, the point in LDS is evaluated again
  LmState& st = sh.st;
  const int fix = prm.fix_length;
#ifdef MVO_LM_STAMPS
  const bool stamp_on = stamp_acc_ != nullptr;
  unsigned long long astamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
  mcc_term = 0.0;
  if (d.take_z) {
    // EvaluateGradientAndJacobian = the system assembled at z
    const double* __restrict__ Hnew = hsel ? st.H : sh.Hz;
    const int fl = fcol(lane < n ? lane : 0, fix);
    hsel ^= 1;  // the buffers swap roles: no copy of the 15 x 15 system
    r.g = lane < n ? sh.gz[fl] : 0.0;
    r.hdiag = lane < n ? Hnew[fl * MVO_MAXC + fl] : 0.0;
    if (d.first) {
      // jacobi_scaling, fixed at iteration 0.  Ceres takes the column norm of J, a sum of squares; here
      // H_ll = G^T W G can come out as -1e-37 for a column the data do not constrain at all (every
      // baseline parallel to u: P v = 0 up to rounding) -- the square root of that must not be NaN.
      r.js = 1.0 / (1.0 + sqrt(fmax(r.hdiag, 0.0)));
    }
    // gradient_max_norm = || x - Plus(x, -g) ||_inf
    if (GMN_READY) {
      r.gmn = sh.gmn_z;
    } else {
      double pv[3], pt;
      frames_plus(lane, M, fix, r.xv, r.xt, r.g, -1.0, pv, pt);
      r.gmn = frames_diff_max(lane, M, fix, r.xv, r.xt, pv, pt);
    }
    d.it_gmn = r.gmn;
  }
  MVO_ASTAMP(15);
  if (d.done) return 0;
  // ---- FinalizeIterationAndCheckIfMinimizerCanContinue, the rest: the record and the tests ----
  lm_trace_row(trace, prm, lane, r.ntrace, d, r.radius);
  ++r.ntrace;
  if (r.iteration >= prm.max_it) {
    lm_terminate(r, 1, 1);
    return 0;
  }
  if (d.it_succ && d.it_gmn <= prm.gtol) {
    lm_terminate(r, 0, 2);
    return 0;
  }
  if (r.radius <= 1e-32) {  // min_trust_region_radius
    lm_terminate(r, 0, 5);
    return 0;
  }
  MVO_ASTAMP(9);
  // ---- next iteration: ComputeTrustRegionStep ----
  ++r.iteration;
  if (!r.reuse_diag) r.dg = fmin(fmax(r.js * r.js * r.hdiag, 1e-6), 1e32);  // min/max_lm_diagonal
  r.reuse_diag = 1;
  // (Js^T Js + D^2) y = Js^T r with Js = J diag(jscale), D^2 = diag / radius; step = -y.
  // Ceres solves the equivalent least-squares problem by dense QR; here the reduced system is
  // eliminated in registers, one row per lane (Gauss-Jordan, no pivoting: the matrix is SPD).
  const double jsm = lane < n ? r.js : 0.0;
  double a[MVO_MAXC];
  {
    const double* __restrict__ Hacc = hsel ? sh.Hz : st.H;
    const int fl = fcol(lane < n ? lane : 0, fix);
    double hrow[MVO_MAXC];
#pragma unroll
    for (int j = 0; j < MVO_MAXC; ++j) hrow[j] = Hacc[fl * MVO_MAXC + (fix ? 3 * (j >> 1) + (j & 1) : j)];  // loads first
    gj_setup<0>(a, hrow, jsm);
  }
  const double dd = r.dg * fast_rcp(r.radius);
  const double ddl = lane < n ? dd : 1.0;  // added to a row's diagonal entry when it becomes the pivot
  const double rhs0 = jsm * r.g;
  MVO_ASTAMP(10);
  double rhs = rhs0, myrinv = 1.0;
  {
    const double piv0 = bcast_lane(a[0] + ddl, 0);
    gj_eliminate<0>(a, rhs, myrinv, lane, ddl, piv0, fast_rcp(piv0), n);
  }
  MVO_ASTAMP(11);
  const double step = lane < n ? -(rhs * myrinv) : 0.0;
  // a pivot < 0 or infinite or NaN (its reciprocal is not > 0), or a non-finite step (a zero pivot gives one): the
  // linear solver failed
  const bool ok = !__any(!(myrinv > 0.0) || !isfinite(step));
  // model_cost_change = -(Js step)^T (r + Js step / 2) = -step^T b - step^T As step / 2 with
  // b = Js^T r, As = Js^T Js.  The step solves (As + D^2) step = -b, so As step = -b - D^2 step
  // and the change is (-step^T b + step^T D^2 step) / 2 -- no second pass over the matrix.
  mcc_term = ok ? step * (dd * step - rhs0) : 0.0;  // (an invalid step: sum 0, which fails the sign test)
  MVO_ASTAMP(12);
  r.phase = PH_EVAL_CAND;
  if (!ok) return 2;
  // ---- ComputeCandidatePoint: delta = step o jscale; candidate = Plus(x, delta) ----
  frames_plus(lane, M, fix, r.xv, r.xt, step * r.js, 1.0, r.cv, r.ct);
  MVO_ASTAMP(13);
  lm_publish_point_early(sh, prm, M, lane, r.cv, r.ct, pub_s, pub_sp);
  MVO_ASTAMP(14);
  return 1;
}

// The rest of the step, off the critical path: the model cost change and the step's validity.  When the step was
// invalid the point in LDS is stale (the last candidate, or x0) and its evaluation will be ignored.
__device__ __forceinline__ void lm_model_cost_change(LmRegs& r, double mcc_term) {
  const double mcc = 0.5 * row0_sum(mcc_term);
  r.mcc = mcc;
  if (mcc > 0.0) r.num_invalid = 0;
  else r.pending_invalid = 1;
}

// ---- stage (4) of an evaluation: the blocks of the system at z ---------------------------------
// 3x3 blocks G_k^T W_kl G_l, one ROW of a block per lane (45 rows fit a wavefront), and the gradient
// entries G_k^T y_k.  W_kl = D[k][l] for l > k.  The pair (k, k) is kept apart: its baseline is s_k v_k
// alone, so r does not depend on t_k at all (IR.cpp:466-471 normalises the scale away) and its scale
// row / column is exactly zero instead of rounding noise of the size of S -- the diagonal W_kk, y_k
// leave class (k, k) out: D[k][k+1] + C1[k-1][k].
struct BlockRole {
  int nB4, k4, l4, i4;  // lane < nB4: row i4 of block (k4, l4); then 3M gradient entries (k4, i4)
};
__device__ __forceinline__ void upper_block(int M, int t, int& k, int& l);
__device__ __forceinline__ int sym6(int r, int c);
__device__ __forceinline__ BlockRole block_role(int M, int lane) {
  BlockRole b;
  b.nB4 = 3 * ((M * (M + 1)) / 2);
  if (lane < b.nB4) {
    b.i4 = lane % 3;
    upper_block(M, lane / 3, b.k4, b.l4);
  } else {
    const int e = lane - b.nB4 < 3 * M ? lane - b.nB4 : 0;
    b.k4 = b.l4 = e / 3;
    b.i4 = e - 3 * b.k4;
  }
  return b;
}

// WHICH: 0 = every block row, 1 = the rows of the diagonal blocks only, 2 = the others only (k_lm<256>
// gives the two kinds to two wavefronts: each then runs a stream without the other's special cases)
template <int WHICH, class SH>
__device__ __forceinline__ void lm_block_rows(SH& sh, const BlockRole& br, int lane, double* __restrict__ Hw) {
  const int k4 = br.k4, l4 = br.l4, i4 = br.i4;
  if (lane < br.nB4 && (WHICH == 0 || (WHICH == 1) == (k4 == l4))) {
    const bool diag = WHICH == 0 ? k4 == l4 : WHICH == 1;
    const double* __restrict__ wsrc = sh.D[k4 * MCX + (diag ? k4 + 1 : l4)];
    const double* __restrict__ csrc = sh.C1[(k4 > 0 ? k4 - 1 : 0) * MCX + k4];
    // every load of the stage up front, unconditionally: a load under a lane's condition is a branch, and each such
    // region is an LDS round trip of its own behind the previous one
    double Wv[6], gk[3], Gl[9], Ts[6];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
      const double cv = csrc[v];
      Wv[v] = wsrc[v] + ((diag && k4 > 0) ? cv : 0.0);
      Ts[v] = (WHICH == 2) ? 0.0 : sh.kk[k4][v];  // (the pair (k, k): only the diagonal blocks use it)
    }
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) gk[rr] = sh.G[k4][3 * rr + i4];
#pragma unroll
    for (int e = 0; e < 9; ++e) Gl[e] = sh.G[l4][e];
    double val[3];
    {
      double gw[3];  // (G_k^T W)[i][m]
#pragma unroll
      for (int m = 0; m < 3; ++m) gw[m] = gk[0] * Wv[sym6(0, m)] + gk[1] * Wv[sym6(1, m)] + gk[2] * Wv[sym6(2, m)];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) val[jj] = gw[0] * Gl[jj] + gw[1] * Gl[3 + jj] + gw[2] * Gl[6 + jj];
    }
    if (diag && i4 < 2) {  // the pair (k,k): direction columns only
      double gw[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) gw[m] = gk[0] * Ts[sym6(0, m)] + gk[1] * Ts[sym6(1, m)] + gk[2] * Ts[sym6(2, m)];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) val[jj] += gw[0] * Gl[jj] + gw[1] * Gl[3 + jj] + gw[2] * Gl[6 + jj];
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
      if (!diag || jj >= i4) {  // diagonal blocks: the upper entry is mirrored, H stays symmetric
        Hw[(3 * k4 + i4) * MVO_MAXC + 3 * l4 + jj] = val[jj];
        Hw[(3 * l4 + jj) * MVO_MAXC + 3 * k4 + i4] = val[jj];
      }
  }
}

// gradient entry (k, i) = G_k^T y_k, for the lanes `on`
template <class SH>
__device__ __forceinline__ void lm_gradient_entry(SH& sh, bool on, int k4, int i4) {
  if (on) {
    const double* __restrict__ dk = sh.D[k4 * MCX + k4 + 1];
    const double* __restrict__ ck = sh.C1[(k4 > 0 ? k4 - 1 : 0) * MCX + k4];
    double gi = 0.0;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const double cv = ck[6 + v], kv = sh.kk[k4][6 + v];  // (loaded by every lane: see lm_block_rows)
      const double yv = dk[6 + v] + (k4 > 0 ? cv : 0.0) + (i4 < 2 ? kv : 0.0);
      gi += sh.G[k4][3 * v + i4] * yv;
    }
    sh.gz[3 * k4 + i4] = gi;
  }
}

// k_lm<256>, a helper wavefront: the gradient at z and its max norm || z - Plus(z, -g) ||_inf, the
// same arithmetic lm_step performs in k_lm<64> (frames_plus with sign = -1, frames_diff_max).
__device__ __forceinline__ void lm_gradient_and_norm(LmShared& sh, LmTables& tb, int M, int fix, int lane) {
  const int e = lane < 3 * M ? lane : 0;
  lm_gradient_entry(tb, lane < 3 * M, e / 3, e % 3);
  wave0_sync();  // this wavefront's own LDS writes, read back below
  const int k = lane < M ? lane : 0;
  const double xv[3] = {sh.zvec[k][0], sh.zvec[k][1], sh.zvec[k][2]};
  const double xt = sh.zt[k];
  const double ga = sh.gz[3 * k], gb = sh.gz[3 * k + 1], gt = sh.gz[3 * k + 2];
  double pv[3];
  d_plus(xv, -1.0 * ga, -1.0 * gb, pv);
  const double pt = fix ? xt : xt + -1.0 * gt;
  const double gmn = frames_diff_max(lane, M, fix, xv, xt, pv, pt);
  if (lane == 0) sh.gmn_z = gmn;
}

// t-th block (k, l), k <= l < M, of the upper triangle in row-major order
__device__ __forceinline__ void upper_block(int M, int t, int& k, int& l) {
  k = 0;
#pragma unroll
  for (int q = 1; q < MVO_MAXM; ++q)
    if (q < M && t >= q * M - (q * (q - 1)) / 2) k = q;
  l = k + (t - (k * M - (k * (k - 1)) / 2));
}

// (a, b) of pair p in the reference's loop order (inverse of pair_index)
__device__ __forceinline__ void pair_ab(int N, int M, int p, int& a, int& b) {
  a = 0;
#pragma unroll
  for (int k = 1; k < MVO_MAXM; ++k)
    if (k < M && p >= pair_index(N, k, k)) a = k;
  b = a + (p - pair_index(N, a, a));
}

struct PairIn {
  double S[7];
  double off[3];
  int a, b;
};

// per-pair inputs: moments (sum of the pair's slots) and the fixed-frame offset n_b v_b (b >= M)
// COHERENT: the sums were stored by other workgroups of THIS launch (k_lm<256, 7>): device-scope loads
template <bool COHERENT = false>
__device__ __forceinline__ void pair_load(PairIn& in, int p, int N, int M, bool unit_slots,
                                          const PairDev* __restrict__ wp, const double* partial,
                                          uint32_t pair_off, const double* __restrict__ fr) {
  if (unit_slots) {
    pair_ab(N, M, p, in.a, in.b);
    const double* ps = partial + (size_t)(pair_off + p) * 8;
#pragma unroll
    for (int v = 0; v < 7; ++v)
      in.S[v] = COHERENT ? __hip_atomic_load(ps + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ps[v];
  } else {
    const PairDev pd = wp[p];
    in.a = (int)pd.a;
    in.b = (int)pd.b;
#pragma unroll
    for (int v = 0; v < 7; ++v) in.S[v] = 0.0;
    for (uint32_t sl = pd.slot_begin; sl < pd.slot_end; ++sl) {
      const double* ps = partial + (size_t)sl * 8;
#pragma unroll
      for (int v = 0; v < 7; ++v) in.S[v] += ps[v];
    }
  }
  in.off[0] = in.off[1] = in.off[2] = 0.0;
  if (in.b >= M) {  // IR.cpp:137-144: only frame b itself
    const double* __restrict__ f = fr + MVO_FRAME_DOUBLES * in.b;
    const double nb = f[12];
    in.off[0] = nb * f[9];
    in.off[1] = nb * f[10];
    in.off[2] = nb * f[11];
  }
}

// the same from the inline pass's LDS sums (k_lm<64>): pair p is row p
__device__ __forceinline__ void pair_load_lds(PairIn& in, int p, int N, int M, const double* lds_sums,
                                              const double* __restrict__ fr) {
  pair_ab(N, M, p, in.a, in.b);
#pragma unroll
  for (int v = 0; v < 7; ++v) in.S[v] = lds_sums[p * 8 + v];
  in.off[0] = in.off[1] = in.off[2] = 0.0;
  if (in.b >= M) {  // IR.cpp:137-144: only frame b itself
    const double* __restrict__ f = fr + MVO_FRAME_DOUBLES * in.b;
    const double nb = f[12];
    in.off[0] = nb * f[9];
    in.off[1] = nb * f[10];
    in.off[2] = nb * f[11];
  }
}

// s_a v_a + ... + s_kmax v_kmax of the published point, summed left to right (IR.cpp:466-577).  All
// five frames are loaded up front (independent LDS reads, one wait) instead of one dependent LDS
// round trip per term.
template <class SH>
__device__ __forceinline__ void free_baseline(const SH& sh, int a, int kmax, double (&beta)[3]) {
  double zs[MVO_MAXM], zv[MVO_MAXM][3];
#pragma unroll
  for (int k = 0; k < MVO_MAXM; ++k) {
    zs[k] = sh.zs[k];
    zv[k][0] = sh.zvec[k][0];
    zv[k][1] = sh.zvec[k][1];
    zv[k][2] = sh.zvec[k][2];
  }
  // frames outside a .. kmax take part with s = 0: their terms are exact zeros, added in front of or behind the
  // sum without changing a bit of it (one select per frame instead of two per frame and component)
#pragma unroll
  for (int k = 0; k < MVO_MAXM; ++k) zs[k] = (k >= a && k <= kmax) ? zs[k] : 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double acc = zs[0] * zv[0][j];
#pragma unroll
    for (int k = 1; k < MVO_MAXM; ++k) acc += zs[k] * zv[k][j];
    beta[j] = acc;
  }
}

// projected moments of one pair at the published point: T = P S P, q = P S u, cost.
// beta = the free frames' part of the pair's baseline (free_baseline); the fixed-frame offset is added here.
__device__ __forceinline__ void pair_eval(const PairIn& in, const double (&beta)[3], int pass_mode, double (&o)[10]) {
  const double bx = beta[0] + in.off[0], by = beta[1] + in.off[1], bz = beta[2] + in.off[2];
  const double inb = fast_rsqrt(bx * bx + by * by + bz * bz), inb2 = inb * inb;
  const double u0 = bx * inb, u1 = by * inb, u2 = bz * inb;
  const double* S = in.S;
  const double Su0 = S[0] * u0 + S[1] * u1 + S[2] * u2;
  const double Su1 = S[1] * u0 + S[3] * u1 + S[4] * u2;
  const double Su2 = S[2] * u0 + S[4] * u1 + S[5] * u2;
  const double uSu = u0 * Su0 + u1 * Su1 + u2 * Su2;
  o[0] = (S[0] - 2.0 * u0 * Su0 + u0 * u0 * uSu) * inb2;
  o[1] = (S[1] - u0 * Su1 - Su0 * u1 + u0 * u1 * uSu) * inb2;
  o[2] = (S[2] - u0 * Su2 - Su0 * u2 + u0 * u2 * uSu) * inb2;
  o[3] = (S[3] - 2.0 * u1 * Su1 + u1 * u1 * uSu) * inb2;
  o[4] = (S[4] - u1 * Su2 - Su1 * u2 + u1 * u2 * uSu) * inb2;
  o[5] = (S[5] - 2.0 * u2 * Su2 + u2 * u2 * uSu) * inb2;
  o[6] = (Su0 - u0 * uSu) * inb;
  o[7] = (Su1 - u1 * uSu) * inb;
  o[8] = (Su2 - u2 * uSu) * inb;
  o[9] = pass_mode ? 0.5 * S[6] : 0.5 * uSu;  // 1/2 sum rho from the pass, or the moment form
}

// the pair's projected moments as the pass's tail left them (project_pair8): ten doubles, 16-byte loads
// COHERENT: the projections were stored by other workgroups of THIS launch (persistent robust solver): device-scope loads
template <bool COHERENT = false>
__device__ __forceinline__ void proj_load(const double* __restrict__ proj, size_t pair, double (&o)[10]) {
  if (COHERENT) {
    const double* q = proj + 10 * pair;
#pragma unroll
    for (int j = 0; j < 10; ++j) o[j] = __hip_atomic_load(q + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const double2* __restrict__ q = reinterpret_cast<const double2*>(proj + 10 * pair);
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const double2 t = q[j];
    o[2 * j] = t.x;
    o[2 * j + 1] = t.y;
  }
}

// symmetric 3x3 stored as (xx,xy,xz,yy,yz,zz)
__device__ __forceinline__ int sym6(int r, int c) {
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  return lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
}

// Inline evaluation (small windows / many windows): the window's own workgroup runs the observation
// pass -- one wavefront per frame pair, straight from the caller's rays -- and leaves the 8 sums
// of every pair in pair_sums, exactly what k_fold/k_pass + k_reduce_slots would have produced.
// MODE 0: second moments only (no loss; done once per solve)
// MODE 1: pass at the published point, w = 1      (assembly_mode = PASSES)
// MODE 2: pass at the published point, Cauchy(0.5)
// LDS_OUT (k_lm<64>): the sums stay in LDS (lds_sums) -- the wavefront that wrote them reads them right
// back, a store -> load round trip through the L2 per evaluation otherwise.
template <int MODE, int NT, bool LDS_OUT = false>
__device__ __forceinline__ void inline_pass(const LmShared& sh, const SolveParams& prm, const PairDev* __restrict__ wp,
                                            int P, int M, const double* __restrict__ fr,
                                            const double* __restrict__ m0, const double* __restrict__ m1,
                                            double* pair_sums, double* lds_sums = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int p = wave; p < P; p += NT / 64) {
    PairDev pd;
    if (prm.small_tab) {  // (uniform) the pair's place in the ray arrays from the kernel arguments, (a, b) by arithmetic
      uint32_t s0 = 0, s1 = 0;
#pragma unroll
      for (int k = 0; k < MVO_SMALL_PAIRS; ++k) {
        s0 = p == k ? prm.small_offs[k] : s0;
        s1 = p == k ? prm.small_offs[k + 1] : s1;
      }
      int a, b;
      pair_ab(prm.N, M, p, a, b);
      pd.a = (uint32_t)a;
      pd.b = (uint32_t)b;
      pd.src_start = s0;
      pd.count = s1 - s0;
      pd.slot_begin = pd.slot_end = 0;
    } else {
      pd = wp[p];
    }
    const double* __restrict__ R0 = fr + MVO_FRAME_DOUBLES * pd.a;        // IR.cpp:172
    const double* __restrict__ R1 = fr + MVO_FRAME_DOUBLES * (pd.b + 1);  // IR.cpp:171
    const double a0 = R0[0], a1 = R0[1], a2 = R0[2], a3 = R0[3], a4 = R0[4], a5 = R0[5], a6 = R0[6], a7 = R0[7],
                 a8 = R0[8];
    const double b0 = R1[0], b1 = R1[1], b2 = R1[2], b3 = R1[3], b4 = R1[4], b5 = R1[5], b6 = R1[6], b7 = R1[7],
                 b8 = R1[8];
    double u[3] = {0.0, 0.0, 0.0}, nb;
    if (MODE != 0) pair_unit_baseline(fr, M, (int)pd.a, (int)pd.b, sh.zvec, sh.zs, u, nb);
    const double* __restrict__ p0 = m0 + 3 * (size_t)pd.src_start;
    const double* __restrict__ p1 = m1 + 3 * (size_t)pd.src_start;
    double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double prod = 1.0;
    int32_t prod_exp = 0;
    for (uint32_t i0 = lane; i0 < pd.count; i0 += 64 * 128) {
      const uint32_t i1 = (pd.count - i0) > 64 * 128 ? i0 + 64 * 128 : pd.count;
#pragma unroll 2
      for (uint32_t i = i0; i < i1; i += 64) {
        const double x0 = p0[3 * i], y0 = p0[3 * i + 1], z0 = p0[3 * i + 2];
        const double x1 = p1[3 * i], y1 = p1[3 * i + 1], z1 = p1[3 * i + 2];
        const double px = a0 * x0 + a1 * y0 + a2 * z0, py = a3 * x0 + a4 * y0 + a5 * z0,
                     pz = a6 * x0 + a7 * y0 + a8 * z0;
        const double qx = b0 * x1 + b1 * y1 + b2 * z1, qy = b3 * x1 + b4 * y1 + b5 * z1,
                     qz = b6 * x1 + b7 * y1 + b8 * z1;
        const double vx = py * qz - pz * qy, vy = pz * qx - px * qz, vz = px * qy - py * qx;
        if (MODE == 0) {
          S[0] += vx * vx;
          S[1] += vx * vy;
          S[2] += vx * vz;
          S[3] += vy * vy;
          S[4] += vy * vz;
          S[5] += vz * vz;
        } else {
          pass_accum<MODE == 2>(vx, vy, vz, u[0], u[1], u[2], S, prod);
        }
      }
      if (MODE == 2) {
        int e;
        prod = frexp(prod, &e);
        prod_exp += e;
      }
    }
    if (MODE == 2) S[6] = 0.25 * (log(prod) + (double)prod_exp * 0.693147180559945309417232121458);
    const double tot = wave_reduce8(S, lane);
    if ((lane & 7) == 0) {
      if (LDS_OUT) lds_sums[p * 8 + reduce8_value_of_lane(lane)] = tot;
      else pair_sums[(size_t)p * 8 + reduce8_value_of_lane(lane)] = tot;
    }
  }
}

// 10 values x 32 lanes -> 10 totals, separately in the two halves of the wavefront (which hold the
// pairs of two different frames).  After the call lane L holds, with rp = (L >> 4) & 1,
//   y  = the total of value 5 rp + 2 ((L >> 2) & 1) + ((L >> 3) & 1)   (identical within 4 lanes)
//   x4 = the total of value 5 rp + 4                                   (identical within 16 lanes)
// of its half.  Same idea as wave_reduce10: every level trades half of the values for half of the lanes.
__device__ __forceinline__ void half_reduce10(const double (&v)[10], int lane, double& y, double& x4) {
  double w5[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) w5[j] = swap_add16(v[j], v[j + 5]);  // even rows: value j, odd rows: value j + 5
  const bool b3 = lane & 8, b2 = lane & 4;
  double x01, x23;
  {
    const double keep = b3 ? w5[1] : w5[0], send = b3 ? w5[0] : w5[1];
    x01 = keep + dpp_mov_f64<0x128>(send);  // row_ror:8 == lane xor 8
  }
  {
    const double keep = b3 ? w5[3] : w5[2], send = b3 ? w5[2] : w5[3];
    x23 = keep + dpp_mov_f64<0x128>(send);
  }
  x4 = w5[4] + dpp_mov_f64<0x128>(w5[4]);
  {
    const double keep = b2 ? x23 : x01, send = b2 ? x01 : x23;
    y = keep + dpp_mov_f64<0x141>(send);  // row_half_mirror: lane i <- lane i ^ 7, the other value of b2
  }
  y += dpp_mov_f64<0xB1>(y);
  y += dpp_mov_f64<0x4E>(y);
  x4 += dpp_mov_f64<0xB1>(x4);
  x4 += dpp_mov_f64<0x4E>(x4);
  x4 += dpp_mov_f64<0x141>(x4);
}

// k_lm<256>: a pair (a, j < M) is a class of its own; the pair (k, k) is kept a second time (see lm_block_rows)
__device__ __forceinline__ void lm_store_short_pair(LmShared& sh, bool on, int a, int j, const double (&o)[10]) {
  if (on) {
    double* __restrict__ dst = sh.cls[a * MCX + j];
#pragma unroll
    for (int v = 0; v < 10; ++v) dst[v] = o[v];
    if (a == j) {
#pragma unroll
      for (int v = 0; v < 10; ++v) sh.kk[a][v] = o[v];
    }
  }
}

// k_lm<256>, helper wavefront cw (0..2) between barriers B1 and B1c: its own copy of the prefix / suffix
// tables, then its share of the system at z -- the diagonal blocks, the gradient and its norm, the other blocks.
__device__ __forceinline__ void lm_helper_products(LmShared& sh, const BlockRole& br, int cw, int M, int N, int fix,
                                                   int lane) {
  LmTables tb = lm_tables(sh, cw);
  lm_prefix_suffix<false>(sh.cls, nullptr, tb.C1, tb.D, M, N, lane);
  if (cw == 0) lm_block_rows<1>(tb, br, lane, sh.hsel ? sh.st.H : sh.Hz);
  else if (cw == 1) lm_gradient_and_norm(sh, tb, M, fix, lane);
  else lm_block_rows<2>(tb, br, lane, sh.hsel ? sh.st.H : sh.Hz);
}

// ---- k_lm<256, SPEC != 0>: the numeric half of the LM step on a wavefront of its own -------------------------
// Wavefront 0 keeps Ceres' control logic (verdict, step evaluator, tolerances, counters, the best point, the
// records); class wavefront 2 -- idle between barriers B1c and B2 like the other two -- keeps the numbers of the
// step: gradient, jacobi scaling, LM diagonal, the accepted point and the candidate, one matrix row per lane.
// Between B1c and B2 it takes the system over if the verdict says so, eliminates, forms the candidate and publishes
// it, while wavefront 0 finishes the iteration's bookkeeping (gradient tolerance, record, counters) beside it.
// What wavefront 0's instruction stream carried around the elimination before -- some forty scalars of state held
// in vector registers, their copies at every join of the control flow, spills to the accumulation registers -- is
// off this path: 5.5 k -> 3.x k cycles from B1c to B2 at config 2.  The step is computed before the control
// wavefront knows whether the minimiser goes on (speculation: when it terminates, the candidate is dropped).
// Hand-over through LDS: verdict (take z / first / done / reuse the diagonal / radius) before B1c; per-row terms of
// the model cost change, solver success and the candidate (zvec, zt, zs, zsp) before B2.
struct LmSolver {
  double g, js, dg, hdiag;     // lane l < n: column l
  double xv[3], xt, cv[3], ct;  // lane k < M: accepted point / candidate of frame k
  int hsel;
};

__device__ __forceinline__ void lm_solver_init(LmSolver& q, const LmShared& sh, int lane) {
  const int k = lane < MVO_MAXM ? lane : 0;
  q.g = q.js = q.dg = q.hdiag = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) q.xv[j] = q.cv[j] = sh.st.x_vec[k][j];
  q.xt = q.ct = sh.st.x_t[k];
  q.hsel = 0;
}

__device__ __forceinline__ void lm_solver_step(LmSolver& q, LmShared& sh, const SolveParams& prm, int M, int n,
                                               int lane) {
  if (sh.ctl_done) return;  // (uniform) the window has terminated: wavefront 0 clears sh.go
  const int take = sh.ctl_take, first = sh.ctl_first, reuse = sh.ctl_reuse;
  const double radius = sh.ctl_radius;
  const int fix = prm.fix_length;
  LmState& st = sh.st;
  if (take) {
    // x <- z; EvaluateGradientAndJacobian = the system assembled at z
    const double* __restrict__ Hnew = q.hsel ? st.H : sh.Hz;
    const int fl = fcol(lane < n ? lane : 0, fix);
    q.hsel ^= 1;
    q.g = lane < n ? sh.gz[fl] : 0.0;
    q.hdiag = lane < n ? Hnew[fl * MVO_MAXC + fl] : 0.0;
    q.xv[0] = q.cv[0]; q.xv[1] = q.cv[1]; q.xv[2] = q.cv[2]; q.xt = q.ct;
    if (first) q.js = 1.0 / (1.0 + sqrt(fmax(q.hdiag, 0.0)));  // jacobi_scaling (see lm_step)
  }
  if (!reuse) q.dg = fmin(fmax(q.js * q.js * q.hdiag, 1e-6), 1e32);  // min/max_lm_diagonal
  const double jsm = lane < n ? q.js : 0.0;
  double a[MVO_MAXC];
  {
    const double* __restrict__ Hacc = q.hsel ? sh.Hz : st.H;
    const int fl = fcol(lane < n ? lane : 0, fix);
    double hrow[MVO_MAXC];
#pragma unroll
    for (int j = 0; j < MVO_MAXC; ++j) hrow[j] = Hacc[fl * MVO_MAXC + (fix ? 3 * (j >> 1) + (j & 1) : j)];
    gj_setup<0>(a, hrow, jsm);
  }
  const double dd = q.dg * fast_rcp(radius);
  const double ddl = lane < n ? dd : 1.0;
  const double rhs0 = jsm * q.g;
  double rhs = rhs0, myrinv = 1.0;
  {
    const double piv0 = bcast_lane(a[0] + ddl, 0);
    if (n == MVO_MAXC) gj_eliminate<0, true>(a, rhs, myrinv, lane, ddl, piv0, fast_rcp(piv0), n);
    else gj_eliminate<0>(a, rhs, myrinv, lane, ddl, piv0, fast_rcp(piv0), n);
  }
  const double step = lane < n ? -(rhs * myrinv) : 0.0;
  const bool ok = !__any(!(myrinv > 0.0) || !isfinite(step));
  if (lane <= MVO_MAXC) sh.mcc_term[lane] = ok ? step * (dd * step - rhs0) : 0.0;  // (see lm_step)
  if (lane == 0) sh.step_ok = ok ? 1 : 0;
  if (!ok) return;
  frames_plus(lane, M, fix, q.xv, q.xt, step * q.js, 1.0, q.cv, q.ct);
  double s_, sp_;
  lm_publish_point_early(sh, prm, M, lane, q.cv, q.ct, s_, sp_);
  if (lane < M) {
    sh.zt[lane] = q.ct;
    sh.zsp[lane] = sp_;
  }
}

// ... and what is left for wavefront 0 between B1c and B2: the rest of FinalizeIterationAndCheckIfMinimizerCanContinue.
// Returns 1 when the minimiser goes on (the solver wavefront's step counts), 0 when the window has terminated.
__device__ __forceinline__ int lm_control_finish(LmRegs& r, LmShared& sh, const SolveParams& prm, int lane,
                                                 TraceRec* __restrict__ trace, int& hsel, LmDecision& d) {
  if (d.take_z) {
    hsel ^= 1;
    r.gmn = sh.gmn_z;
    d.it_gmn = r.gmn;
  }
  if (d.done) return 0;
  lm_trace_row(trace, prm, lane, r.ntrace, d, r.radius);
  ++r.ntrace;
  if (r.iteration >= prm.max_it) {
    lm_terminate(r, 1, 1);
    return 0;
  }
  if (d.it_succ && d.it_gmn <= prm.gtol) {
    lm_terminate(r, 0, 2);
    return 0;
  }
  if (r.radius <= 1e-32) {  // min_trust_region_radius
    lm_terminate(r, 0, 5);
    return 0;
  }
  ++r.iteration;
  r.reuse_diag = 1;
  r.phase = PH_EVAL_CAND;
  return 1;
}

// k_lm<256>, wavefronts 1..3: the loop of a class wavefront.  Two layouts, chosen per window:
//   N - 1 - M <= 32 (HALF): wavefront w evaluates the pairs (a, b), a = 2 (w - 1) + (lane >> 5),
//                   b = M + (lane & 31): one pair per lane, one evaluation, one reduction per half;
//   otherwise:      a in {w - 1, w + 2}, b = M + lane + 64 r (r < 4), added up per frame.
// Either way the class totals end up in sh.cls[a][M].  Then, while wavefront 0 decides on the point
// (lm_decide), wavefronts 1 and 3 multiply out the diagonal / the other blocks of the system and
// wavefront 2 the gradient and its norm.  A class wavefront executes exactly the workgroup barriers wavefront 0 executes (marked
// B0 / B1 / B1c / B2 on both sides).
template <int NT, bool SPLIT, bool HALF_ONLY, bool SHORT_HERE, bool COHERENT_PROJ = false>
__device__ __forceinline__ void lm_class_wave(LmShared& sh, const SolveParams& prm, int cw, int lane, int M, int N, int P,
                                              bool& need_pass, int unit_slots, const PairDev* __restrict__ wp,
                                              const double* partial, uint32_t pair_off, const double* __restrict__ fr,
                                              const double* __restrict__ m0, const double* __restrict__ m1,
                                              double* my_sums, const double* proj) {
  if (prm.proj_ready) {
    // Pass mode with the passes as launches of their own: this launch is ONE evaluation long and the pass's
    // tails left every pair projected (project_pair8, mvo_kernels.h).  Wavefront cw adds up the long classes
    // of frames cw and cw + 3: lanes over b, all loads of both frames requested before the first use (one
    // memory latency), then one register butterfly per frame.  Same barriers as the loop below.
    constexpr int ROUNDS = 4;
    const int n_pairs_a = N - 1 > M ? N - 1 - M : 0;
    const int n_rounds = (n_pairs_a + 63) >> 6;  // wave-uniform
    const double* pw = proj + 10 * (size_t)pair_off;
    double o[2][ROUNDS][10];
    bool ok[2][ROUNDS];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int rr = 0; rr < ROUNDS; ++rr) {
        const int a = cw + 3 * q, b = M + lane + 64 * rr;
        ok[q][rr] = a < M && b <= N - 2;
        if (rr < n_rounds && (q == 0 || cw + 3 < M))  // wave-uniform: rounds and frames that exist
          proj_load<COHERENT_PROJ>(pw, (size_t)(ok[q][rr] ? pair_index(N, a, b) : 0), o[q][rr]);
      }
    const BlockRole br = block_role(M, lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int a = cw + 3 * q;
      if (a < M) {  // wave-uniform
        double acc[10];
#pragma unroll
        for (int v = 0; v < 10; ++v) {
          acc[v] = 0.0;
#pragma unroll
          for (int rr = 0; rr < ROUNDS; ++rr)
            if (rr < n_rounds) acc[v] += ok[q][rr] ? o[q][rr][v] : 0.0;
        }
        double t8, t2;
        wave_reduce10(acc, lane, t8, t2);
        if ((lane & 7) == 0) sh.cls[a * MCX + M][reduce8_value_of_lane(lane)] = t8;
        if ((lane & 31) == 0) sh.cls[a * MCX + M][8 + (lane >> 5)] = t2;
      }
    }
    __syncthreads();  // B1
    lm_helper_products(sh, br, cw, M, N, prm.fix_length, lane);
    __syncthreads();  // B1c
    __syncthreads();  // B2
    if (sh.go) need_pass = true;
    return;
  }
  if (prm.cls_ready) {
    // Observation-partitioned pass mode: the class totals arrive ready-made, summed over all ranks
    // (k_class_sums + all-reduce), and this launch is one evaluation long -- the class wavefronts only
    // multiply out the system.  Same barriers as below.
    const BlockRole br = block_role(M, lane);
    __syncthreads();  // B1
    lm_helper_products(sh, br, cw, M, N, prm.fix_length, lane);
    __syncthreads();  // B1c
    __syncthreads();  // B2
    if (sh.go) need_pass = true;
    return;
  }
  constexpr int ROUNDS = 4;  // class (a, M) has at most MVO_HIP_MAX_FRAMES - 1 - M <= 256 pairs
  constexpr int NA = 2;      // frames per class wavefront: cw and cw + 3
  const int n_pairs_a = N - 1 > M ? N - 1 - M : 0;
  const bool half = HALF_ONLY || n_pairs_a <= 32;  // wave-uniform (HALF_ONLY: the host checked, lm_spec)
  const int n_rounds = (n_pairs_a + 63) >> 6;  // wave-uniform
  const bool two = cw + 3 < M;                 // wave-uniform: a second frame to serve
  int my_p[NA][ROUNDS];
  bool my_ok[NA][ROUNDS];
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int rr = 0; rr < ROUNDS; ++rr) {
      const int a = cw + 3 * q, b = M + lane + 64 * rr;
      my_ok[q][rr] = a < M && b <= N - 2;
      my_p[q][rr] = my_ok[q][rr] ? pair_index(N, a, b) : 0;
    }
  const int a_h = 2 * cw + (lane >> 5), b_h = M + (lane & 31);
  if (half) {
    my_ok[0][0] = a_h < M && b_h <= N - 2;
    my_p[0][0] = my_ok[0][0] ? pair_index(N, a_h, b_h) : 0;
  }
  // SHORT_HERE (HALF layout only): the <= 15 pairs (a, b < M), each a class of its own, ride along in the upper
  // half of class wavefront 2, whose frame there (a = 5) never exists -- wavefront 0 has nothing to evaluate and
  // spends the time behind barrier B2 on the rest of its bookkeeping.
  const bool short_wave = SHORT_HERE && cw == 2;  // wave-uniform
  const int nshort = (M * (M + 1)) / 2;
  const bool is_short = short_wave && lane >= 32 && lane - 32 < nshort;
  int s_a = 0, s_j = 0;
  upper_block(M, is_short ? lane - 32 : 0, s_a, s_j);  // (a, b) with a <= b < M, row-major
  if (is_short) my_p[0][0] = pair_index(N, s_a, s_j);  // (my_ok stays false: not part of the half's class total)
  const BlockRole br = block_role(M, lane);
  PairIn in[NA][ROUNDS];
  bool have_in = false;
  LmSolver slv;
  const bool solver_wave = SPLIT && cw == 2;  // wave-uniform
  const int n_cols = prm.fix_length ? 2 * M : 3 * M;
  if (solver_wave) lm_solver_init(slv, sh, lane);
  while (!need_pass) {
    if (prm.inline_eval && prm.pass_mode) {
      if (prm.use_loss) inline_pass<2, NT>(sh, prm, wp, P, M, fr, m0, m1, my_sums);
      else inline_pass<1, NT>(sh, prm, wp, P, M, fr, m0, m1, my_sums);
      __syncthreads();  // B0
    }
    if (!have_in) {
#pragma unroll
      for (int q = 0; q < NA; ++q)
        if (q == 0 || (two && !half)) {
          if (n_rounds <= 1) {
            pair_load<COHERENT_PROJ>(in[q][0], my_p[q][0], N, M, unit_slots != 0, wp, partial, pair_off, fr);
          } else {
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr)
              pair_load<COHERENT_PROJ>(in[q][rr], my_p[q][rr], N, M, unit_slots != 0, wp, partial, pair_off, fr);
          }
        }
      have_in = !prm.pass_mode;
    }
    if (half) {
      double beta[3], o[10];
      free_baseline(sh, is_short ? s_a : (a_h < M ? a_h : 0), is_short ? s_j : M - 1, beta);
      pair_eval(in[0][0], beta, prm.pass_mode, o);
      if (short_wave) lm_store_short_pair(sh, is_short, s_a, s_j, o);
#pragma unroll
      for (int v = 0; v < 10; ++v) o[v] = my_ok[0][0] ? o[v] : 0.0;
      double y, x4;
      half_reduce10(o, lane, y, x4);
      if (a_h < M) {
        const int rp5 = 5 * ((lane >> 4) & 1);
        if ((lane & 3) == 0) sh.cls[a_h * MCX + M][rp5 + ((lane >> 1) & 2) + ((lane >> 3) & 1)] = y;
        if ((lane & 15) == 0) sh.cls[a_h * MCX + M][rp5 + 4] = x4;
      }
    } else {
      // every pair of frame a has the free frames a .. M-1: one baseline per frame and evaluation; the
      // slots of one lane are independent chains (sqrt, reciprocal, ~100 flops), issued interleaved
      double acc[NA][10];
#pragma unroll
      for (int q = 0; q < NA; ++q) {
#pragma unroll
        for (int v = 0; v < 10; ++v) acc[q][v] = 0.0;
        if (q == 0 || two) {
          const int a = cw + 3 * q;
          double beta[3];
          free_baseline(sh, a < M ? a : 0, M - 1, beta);
          if (n_rounds <= 1) {
            double o[10];
            pair_eval(in[q][0], beta, prm.pass_mode, o);
#pragma unroll
            for (int v = 0; v < 10; ++v) acc[q][v] = my_ok[q][0] ? o[v] : 0.0;
          } else {
            double o[ROUNDS][10];
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr) pair_eval(in[q][rr], beta, prm.pass_mode, o[rr]);
#pragma unroll
            for (int v = 0; v < 10; ++v) {
              acc[q][v] = my_ok[q][0] ? o[0][v] : 0.0;
#pragma unroll
              for (int rr = 1; rr < ROUNDS; ++rr) acc[q][v] += my_ok[q][rr] ? o[rr][v] : 0.0;
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NA; ++q)
        if (q == 0 || two) {
          const int a = cw + 3 * q;
          double t8, t2;
          wave_reduce10(acc[q], lane, t8, t2);
          if (a < M) {
            if ((lane & 7) == 0) sh.cls[a * MCX + M][reduce8_value_of_lane(lane)] = t8;
            if ((lane & 31) == 0) sh.cls[a * MCX + M][8 + (lane >> 5)] = t2;
          }
        }
    }
    __syncthreads();  // B1: class totals are in LDS
    lm_helper_products(sh, br, cw, M, N, prm.fix_length, lane);
    __syncthreads();  // B1c: the system at z is in LDS
    if (solver_wave) lm_solver_step(slv, sh, prm, M, n_cols, lane);
    __syncthreads();  // B2: the step is taken and the next point published
    if (!sh.go) break;
    if (prm.pass_mode && !prm.inline_eval) need_pass = true;
  }
}

// ---- k_lm<256, 6>: the robust path of ONE window as ONE persistent launch -----------------------------------------
// (IterativeRefinement.cpp:85-88 + :231-233: ceres::Solve with a loss function re-weights every residual at every
// evaluation.)  The launch-per-evaluation path pays, per evaluation, a pass launch, a k_lm relaunch (state in and out
// of HBM, ~5 of its ~8 us) and two launch boundaries.  Here the grid of the PASS (one workgroup per four wave-chunks of
// residual blocks, all resident, at most one per CU) stays for the whole solve, and EVERY workgroup runs the LM step
// itself -- replicated: the same instruction stream on the same numbers, bit for bit the same verdicts and candidates
// everywhere -- so that the step has nobody to hand its result to.  Per evaluation:
//   1. every workgroup: class totals from the pairs' projections (device-scope loads), system, verdict, step, candidate
//      (the SPEC-5 evaluation, state in registers for the whole solve);
//   2. every workgroup: unit baselines of ALL pairs at the candidate into its LDS (the epilogue's arithmetic);
//   3. the pass over the workgroup's own tiles, slots stored device-coherently, per-pair tickets, the workgroup that
//      completes a pair adds its slots up and projects it (block_finish_pairs) into the OTHER projection buffer;
//   4. grid_sync (mvo_grid.h) -- the only wait, bounded, with a give-up flag.
// Two projection buffers alternate, so a workgroup that is one evaluation ahead never writes what another still reads;
// slots and tickets are rewritten only behind the barrier.  Workgroup 0 writes the result.  If a wait gives up the
// launch ends without a result, the host sees the error word and repeats the solve launch by launch.
// Eligible (host, robust_persistent_ok): one window, pass workgroups <= kRobustMaxWorkgroups, pairs <=
// kRobustMaxPairs, tiles per workgroup <= kRobustMaxTiles, the SPEC-5 conditions (all columns free, no trace).
constexpr int kRobustMaxWorkgroups = 192, kRobustMaxPairs = 512, kRobustMaxTiles = 64;
struct RobustArgs {
  const TileDev* tiles;
  const uint32_t* wave_tiles;
  const double* cbuf;
  size_t cstride;
  double* partial;
  int32_t* pair_ticket;
  const PairDev* pairs_slots;  // the pairs with their slot ranges (k_lm's own `pairs` are the unit rows here)
  double* pair_sums;
  double* proj[2];             // [0]: what the fold's tails wrote (evaluation 0)
  GridSync gs;
  uint32_t n_waves, n_wg;
  int32_t robust;              // Cauchy(0.5) / plain (assembly_mode = PASSES)
  // SPEC 7 (fold and LM loop in one launch): gs.count counts finished pairs and is never reset -- workgroup 0 goes on when
  // it has reached `target` (the host knows what it was before the launch and how many pairs have residual blocks)
  uint32_t target, spin_limit;
  // SPEC 6: the counter the NEXT persistent launch on this handle will use (the launches alternate between two): zeroed
  // by this launch, so that no clearing command stands in front of any of them
  uint32_t* next_count;
};

// SPEC: the same kernel with the run-time switches of the common solves fixed at compile time, so that the
// compiler sees one path instead of all of them (fewer live values around wavefront 0's loop, fewer hoisted
// addresses spilled to lanes and read back):
//   0  every switch at run time (loss / pass modes, fixLength, traces, partitioned solves)
//   1  no loss, observations folded by the chip-wide kernels, all columns free, no trace  (one large window)
//   2  the same with the window's own workgroup folding its observations                   (small windows)
//   3, 4  = 1, 2 for windows with at most 32 pairs (a, b >= M) per refined frame (N - 1 - M <= 32): k_lm<256> hands
//      the numeric half of the LM step to a class wavefront (LmSolver); k_lm<64> is the same kernel as 1, 2
//   5  the robust path's relaunch: one evaluation per launch from the pairs the pass's tails projected (pass_mode,
//      x0_ready, proj_ready; with or without the loss), all columns free, no trace
//   6  = 5 as ONE persistent launch over the pass's grid (RobustArgs above)
//   7  = 3 with the fold in the same launch: workgroups 1 .. n are the FOLD's grid, each folds its tiles and leaves
//      through the ticket tail as in k_fold; workgroup 0 sets the LM loop up, waits until every pair is finished (a
//      count of finished pairs, bounded polling, nobody waits for workgroup 0) and runs it.  One small window whose fold does not need
//      more than a workgroup per CU (config 2: 118): saves the boundary between two dependent launches
// The host picks it from SolveParams (lm_spec, mvo_hip.hip); MVO_HIP_LM_GENERIC=1 keeps everything on 0.
template <int NT, int SPEC = 0>
__global__ __launch_bounds__(NT) void k_lm(SolveParams prm, int unit_slots, const WinDev* __restrict__ wins,
                                           const double* __restrict__ frames, const double* __restrict__ init_x,
                                           const PairDev* __restrict__ pairs, const double* partial,
                                           const double* __restrict__ m0, const double* __restrict__ m1,
                                           double* pair_sums,  // may alias `partial` (inline evaluation)
                                           LmState* __restrict__ states,
                                           LmResult* __restrict__ results,
                                           int32_t* __restrict__ win_status,
                                           TraceRec* __restrict__ trace, int32_t* __restrict__ done_count,
                                           const double* proj,                 // prm.proj_ready: 10 doubles per pair
                                           const double* __restrict__ cls_in,  // prm.cls_ready: MVO_CLS_DOUBLES per window
                                           double* __restrict__ pair_u,        // pass mode: (u, |baseline|) per pair for the next pass
                                           RobustArgs ra) {                    // SPEC 6 only
  constexpr bool LARGE = NT > 64;
  static_assert(NT == 64 || NT == MVO_LM_NT_LARGE, "one wavefront, or three class wavefronts + one for the short pairs and the LM step");
  if constexpr (SPEC != 0) {
    prm.cls_ready = 0;
    prm.fix_length = 0;
    prm.trace_cap = 0;
    trace = nullptr;
    if constexpr (SPEC == 5 || SPEC == 6) {
      prm.pass_mode = 1;
      prm.inline_eval = 0;
      prm.x0_ready = 1;
      prm.proj_ready = 1;
    } else {
      prm.use_loss = 0;
      prm.pass_mode = 0;
      prm.inline_eval = (SPEC == 2 || SPEC == 4) ? 1 : 0;
      prm.x0_ready = 0;
      prm.proj_ready = 0;
    }
  }
  __shared__ LmShared sh;
  __shared__ double pairbuf[LARGE ? 1 : 64 * 10];  // k_lm<64>: P x 10: T(6) q(3) cost(1)
  __shared__ double inl_sums[LARGE ? 1 : 64 * 8];   // k_lm<64>, inline evaluation: the passes' 8 sums per pair
  // pass mode with the passes as launches of their own: the fixed-frame offsets n_b v_b (IR.cpp:137-144) that
  // the epilogue's unit baselines need, requested in the launch's first instructions together with the state
  constexpr int FOFF_FRAMES = LARGE ? MVO_HIP_MAX_FRAMES : 72;  // k_lm<64>: P <= 64, hence N <= 65
  __shared__ double sfoff[3 * FOFF_FRAMES];
  constexpr bool PERSIST = SPEC == 6;  // every workgroup of the grid solves window 0 (replicated) and passes over its own tiles
  static_assert(!PERSIST || NT == FOLD_THREADS, "the persistent solver runs the pass with its own workgroup");
  __shared__ double su[PERSIST ? 4 * kRobustMaxPairs : 1];   // (u, |baseline|) of every pair at the candidate
  __shared__ double red_buf[(PERSIST || SPEC == 7) ? 8 * kRobustMaxTiles : 1];
  __shared__ TileDev s_tiles[PERSIST ? kRobustMaxTiles : 1];  // this workgroup's tiles and its wavefronts' first tiles:
  __shared__ uint32_t s_wt[FOLD_WAVES + 1];                   // read once, not once per pass (two dependent L2 round trips)
  if (PERSIST && (gridDim.x != ra.n_wg || ra.n_wg > (uint32_t)kRobustMaxWorkgroups)) return;  // (host bug guard: nobody waits yet)
  if (PERSIST && blockIdx.x == 0 && threadIdx.x == 0 && ra.next_count != nullptr)  // (the launch before this one used it and is over)
    __hip_atomic_store(ra.next_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (PERSIST) {  // (read behind the prologue's barriers)
    const uint32_t bw0 = blockIdx.x * FOLD_WAVES;
    const uint32_t bw1 = bw0 + FOLD_WAVES < ra.n_waves ? bw0 + FOLD_WAVES : ra.n_waves;
    const uint32_t bt0 = ra.wave_tiles[bw0], bt1 = ra.wave_tiles[bw1];
    if (bt1 - bt0 > (uint32_t)kRobustMaxTiles) return;  // (host bug guard; uniform over the grid only if the host checked)
    if (threadIdx.x <= FOLD_WAVES) s_wt[threadIdx.x] = ra.wave_tiles[bw0 + threadIdx.x <= bw1 ? bw0 + threadIdx.x : bw1];
    for (uint32_t t = threadIdx.x; t < bt1 - bt0; t += NT) s_tiles[t] = ra.tiles[bt0 + t];
  }
  constexpr bool FUSED = SPEC == 7;
  static_assert(!FUSED || NT == FOLD_THREADS, "the fused launch folds with its own workgroups");
  if (FUSED && blockIdx.x != 0) {
    // ---- the fold, as k_fold<false, true, 0> runs it: this workgroup's tiles, slots out, ticket tail ----
    // (workgroup 0 folds nothing: it is dispatched first and has the LM loop's prologue behind it -- state, starting
    // point, the first published point -- by the time the fold workgroups 1 .. n are through)
    const uint32_t bw0 = (blockIdx.x - 1) * FOLD_WAVES;
    const uint32_t bw1 = bw0 + FOLD_WAVES < ra.n_waves ? bw0 + FOLD_WAVES : ra.n_waves;
    const uint32_t bt0 = ra.wave_tiles[bw0], bt1 = ra.wave_tiles[bw1];
    // (host bug guards; a workgroup that leaves here never reports its pairs: workgroup 0 gives up, the host falls back)
    if (gridDim.x != ra.n_wg + 1 || bt1 - bt0 > (uint32_t)kRobustMaxTiles) return;
    const uint32_t fwv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PassOut po;
    po.upairs = nullptr;
    po.proj = nullptr;
    po.coherent_proj = 0;
    po.coherent_sums = 1;  // read by workgroup 0 of this launch, maybe on another XCD
    fold_wave_tiles<false, true, 0>(ra.tiles, ra.wave_tiles, bw0 + fwv, ra.n_waves, bt0, frames, m0, m1, nullptr, 0, red_buf,
                                    (int)(threadIdx.x & 63), po);
    block_store_partials(ra.tiles, bt0, bt1, red_buf, ra.partial, true);
    const uint32_t nq = block_finish_pairs(ra.pair_ticket, ra.tiles, bt0, bt1, ra.pairs_slots, red_buf, ra.partial,
                                           ra.pair_sums, po);
    // the sums of the pairs this workgroup finished are out (every wavefront waits for its own stores) before they count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && nq) __hip_atomic_fetch_add(ra.gs.count, nq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (FUSED && gridDim.x != ra.n_wg + 1) return;  // (host bug guard, as in the fold workgroups: nobody reports, the host falls back)
  const int w = (PERSIST || FUSED) ? 0 : (int)blockIdx.x + prm.win_base, tid = threadIdx.x;
  const bool lead_wg = !PERSIST || blockIdx.x == 0;  // writes status words and the result
#ifdef MVO_LM_STAMPS
  const unsigned long long stamp_entry_ = __builtin_amdgcn_s_memtime();
#endif
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool wave0 = wv == 0;
  LmState* __restrict__ gst = states + w;
  const bool fresh = prm.fresh != 0;
  if (fresh && tid == 0 && lead_wg) win_status[w] = 0;  // read by k_pass: set again when this window terminates
  // every window of an upload has the same shape (SolveParams): no table look-up in front of everything,
  // and the status word comes in with the state image instead of a round trip of its own (a relaunch of
  // the robust path is one evaluation long: -1.4 k of 7 k prologue cycles at config 4r)
  WinDev win;
  win.frame_off = (uint32_t)w * (uint32_t)prm.N;
  win.pair_off = (uint32_t)w * (uint32_t)prm.P;
  win.N = (uint32_t)prm.N;
  win.M = (uint32_t)prm.M;
  win.P = (uint32_t)prm.P;
  const int M = win.M, N = win.N, P = win.P;
  const int fix = prm.fix_length;
  const int n = fix ? 2 * M : 3 * M;
  const double* __restrict__ fr = frames + (size_t)MVO_FRAME_DOUBLES * win.frame_off;
  const PairDev* __restrict__ wp = pairs + win.pair_off;

  // solver state HBM -> LDS (all threads), then the point to evaluate is published by wave 0
  for (int i = tid; i < (int)(sizeof(LmState) / 8); i += NT)
    reinterpret_cast<double*>(&sh.st)[i] = fresh ? 0.0 : reinterpret_cast<const double*>(gst)[i];
  // The elimination reads whole 15-entry rows of both system buffers; an evaluation writes only the
  // 3M x 3M corner.  LDS keeps whatever the last workgroup on this CU left there -- a NaN pattern in
  // an unused column would poison the right-hand side (finite leftovers are harmless: padding rows
  // are identity rows) -- so the second buffer starts from zero like the first.
  for (int i = tid; i < MVO_MAXC * MVO_MAXC; i += NT) sh.Hz[i] = 0.0;
  if (tid == 0) sh.hsel = 0;
  // frames k >= M are never published; free_baseline multiplies them by s = 0, so they must hold numbers
  if (tid < MVO_MAXM) {
    sh.zs[tid] = 0.0;
    sh.zvec[tid][0] = sh.zvec[tid][1] = sh.zvec[tid][2] = 0.0;
  }
  if (prm.pass_mode && !prm.inline_eval)  // (uniform) this launch ends by publishing every pair's unit baseline
    for (int b = tid; b < N && b < FOFF_FRAMES; b += NT) {
      const double* __restrict__ f = fr + MVO_FRAME_DOUBLES * b;
      const double nb = f[12];
      sfoff[3 * b] = nb * f[9];
      sfoff[3 * b + 1] = nb * f[10];
      sfoff[3 * b + 2] = nb * f[11];
    }
  __syncthreads();
  if (sh.st.status != 0) return;  // a window that has terminated (uniform)
  LmRegs r;      // wavefront 0: the solver state, in registers until the launch ends
  int hsel = 0;  // which of {st.H, Hz} holds the ACCEPTED system; between launches it is st.H
  if (wave0) {
    LmState& st = sh.st;
    const int k = lane < M ? lane : 0;
    if (st.phase == PH_START && lane < M) {
      // IR.cpp:95-109: vec_k from the frame, t_k = reverseScale(scale_k) (host libm, uploaded)
      st.x_vec[k][0] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 0];
      st.x_vec[k][1] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 1];
      st.x_vec[k][2] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 2];
      st.x_t[k] = init_x[((size_t)w * MVO_MAXM + k) * 4 + 3];
    }
    if (st.phase != PH_EVAL_CAND) lm_publish_point(sh, prm, M, lane, st.x_vec[k], st.x_t[k]);
    else lm_publish_point(sh, prm, M, lane, st.c_vec[k], st.c_t[k]);
    if (lane == 0) sh.phase = st.phase;
    wave0_sync();
    regs_load(r, st, fix, n, lane);
  }
  __syncthreads();

#ifdef MVO_LM_STAMPS
  unsigned long long stamp_prev_ = 0;
  __shared__ unsigned long long stamp_lds_[16];
  if (tid < 16) stamp_lds_[tid] = 0;
  __syncthreads();
  unsigned long long* stamp_acc_ = (w == 0) ? stamp_lds_ : nullptr;
  if (w == 0 && tid == 0) {  // slot 6: kernel entry -> here (state in, point published)
    stamp_prev_ = __builtin_amdgcn_s_memtime();
    stamp_lds_[6] = stamp_prev_ - stamp_entry_;
  }
#endif
  if (FUSED) {
    // every pair's sums are complete once the count of finished pairs has reached the target.  Bounded: a fold workgroup
    // that never reports (it cannot: nobody waits but this workgroup) ends the wait with the error word raised and no
    // result, and the host repeats the solve as two launches.
    __shared__ uint32_t fused_ok_s;
    if (tid == 0) {
      uint32_t ok = 1;
      for (uint32_t spins = 0;; ++spins) {
        const uint32_t c = __hip_atomic_load(ra.gs.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(c - ra.target) >= 0) break;
        if (spins >= ra.spin_limit) {
          __hip_atomic_store(ra.gs.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      fused_ok_s = ok;
    }
    __syncthreads();
    if (!fused_ok_s) return;
  }
  bool need_pass = false;
  if (sh.phase == PH_START) {
    r.phase = PH_EVAL_X0;
    // x0_ready: the fold was the pass at the starting point (k_fold<.., PASS0>): its sums are there
    need_pass = prm.pass_mode != 0 && !prm.inline_eval && !prm.x0_ready;
  }
  double* my_sums = pair_sums + (size_t)win.pair_off * 8;
  if (prm.inline_eval && !prm.pass_mode) {  // no loss: the moments are folded once, here
    inline_pass<0, NT, !LARGE>(sh, prm, wp, P, M, fr, m0, m1, my_sums, inl_sums);
    __syncthreads();
  }

  // ---- who evaluates which frame pair (fixed for the launch) ---------------------------------
  // k_lm<64>:  lane p evaluates pair p.
  // k_lm<256>: wavefronts 1..3 take the pairs (a, b >= M) (lm_class_wave); wavefront 0's lane i the
  //            i-th pair with b < M (a = 0: b = 0..M-1, a = 1: b = 1..M-1, ...), then the LM step.
  // SPEC 3 / 4: at most 32 pairs (a, b >= M) per frame -- the class wavefronts hold one pair per lane (HALF layout)
  // and class wavefront 2 has the registers to take the numeric half of the step (LmSolver)
  constexpr bool SPLIT = LARGE && (SPEC == 3 || SPEC == 4 || SPEC == 7);
  // PERSIST: the evaluations of the whole solve, one per turn of this loop (everything else: one turn)
  uint32_t grid_epoch = 0;
  int par = 0;             // which projection buffer holds the evaluation to take in
  bool grid_fail = false;
  if (PERSIST) proj = ra.proj[0];
#ifdef MVO_LM_STAMPS
  unsigned long long rstamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
  for (int turn = 0;; ++turn) {
  if (LARGE && !wave0) lm_class_wave<NT, SPLIT, SPLIT, SPLIT, PERSIST || FUSED>(sh, prm, wv - 1, lane, M, N, P, need_pass, unit_slots, wp, partial, win.pair_off, fr, m0, m1, my_sums, proj);
  int my_p0 = lane < P ? lane : 0;
  bool my_ok0 = lane < P;
  int my_a = 0, my_j = 0;  // k_lm<256>: the short pair's (a, b)
  if (LARGE) {
    const int nshort = (M * (M + 1)) / 2;
    upper_block(M, lane < nshort ? lane : 0, my_a, my_j);  // (a, b) with a <= b < M, row-major: the same enumeration
    my_ok0 = lane < nshort;
    my_p0 = pair_index(N, my_a, my_j);
  }
  PairIn in0;
  bool have_in = false;
  // lane roles of the three assembly stages (fixed): (j, v) of the prefix stage, (k, v) of the suffix
  // stage, block row (k, l, i) / gradient entry (k, i) of the product stage
  const BlockRole br = block_role(M, lane);  // block rows of the upper triangle; then 3M gradient entries

  while (!need_pass && (!LARGE || wave0)) {
    MVO_STAMP(0);
    // ---------------- evaluation at z: assemble J^T J, J^T r, cost ----------------
    if (prm.inline_eval && prm.pass_mode) {  // one observation pass per evaluation, in this workgroup
      if (prm.use_loss) inline_pass<2, NT, !LARGE>(sh, prm, wp, P, M, fr, m0, m1, my_sums, inl_sums);
      else inline_pass<1, NT, !LARGE>(sh, prm, wp, P, M, fr, m0, m1, my_sums, inl_sums);
      __syncthreads();  // B0
    }
    // (1) per frame pair: projected moments.  Without a loss the moments never change: the inputs are
    // loaded once and stay in registers.  Loads are unconditional, at a clamped pair index: a load
    // inside a divergent `if` is waited for on the spot, which would serialise the L2 latencies.
    if (prm.cls_ready) {
      // observation-partitioned pass mode: the class totals of this evaluation arrive ready-made, summed over
      // all ranks (k_class_sums + all-reduce); cls and kk are contiguous in LmShared
      double* __restrict__ dst = &sh.cls[0][0];
      const double2* __restrict__ src = reinterpret_cast<const double2*>(cls_in + (size_t)w * MVO_CLS_DOUBLES);
      constexpr int NV = (MVO_MAXM * MCX + MVO_MAXM) * 10 / 2;  // 175 16-byte pieces: three loads per lane, issued together
      double2 t[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) t[q] = src[lane + 64 * q < NV ? lane + 64 * q : 0];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (lane + 64 * q < NV) {  // (8-byte LDS stores: LmShared::cls is not 16-byte aligned)
          dst[2 * (lane + 64 * q)] = t[q].x;
          dst[2 * (lane + 64 * q) + 1] = t[q].y;
        }
    } else if (SPLIT) {
      // class wavefront 2 evaluates the short pairs as well (lm_class_wave, SHORT_HERE): nothing for this wavefront
    } else {
      double o[10], beta[3];
      int pa, pb;
      if (prm.proj_ready) {
        // pass mode, passes as launches of their own: the pass's tail projected every pair -- nothing to evaluate
        pair_ab(N, M, my_p0, pa, pb);
        proj_load<PERSIST>(proj, (size_t)win.pair_off + my_p0, o);
      } else {
        if (!have_in) {
          if (!LARGE && prm.inline_eval) pair_load_lds(in0, my_p0, N, M, inl_sums, fr);
          else pair_load(in0, my_p0, N, M, unit_slots != 0, wp, partial, win.pair_off, fr);
          have_in = !prm.pass_mode;
        }
        free_baseline(sh, in0.a, in0.b < M - 1 ? in0.b : M - 1, beta);
        pair_eval(in0, beta, prm.pass_mode, o);
        pa = in0.a;
        pb = in0.b;
      }
      if (my_ok0) {
        double* __restrict__ dst = LARGE ? sh.cls[my_a * MCX + my_j] : pairbuf + lane * 10;
#pragma unroll
        for (int v = 0; v < 10; ++v) dst[v] = o[v];
        if (pa == pb) {
#pragma unroll
          for (int v = 0; v < 10; ++v) sh.kk[pa][v] = o[v];
        }
      }
    }
    if (LARGE) __syncthreads();  // B1: the class wavefronts' totals are in LDS
    else wave0_sync();
    MVO_STAMP(1);
    {
      // (2)-(4): prefix / suffix tables, the blocks of the system at z, and the verdict on z.  k_lm<256>: the
      // helper wavefronts form the tables (each its own copy) and multiply out blocks and gradient
      // (lm_helper_products) while this one decides -- the verdict needs the cost only, which it adds up
      // itself; k_lm<64>: one after the other.
      LmDecision dec;
      if (LARGE) {
        const double cost_raw = lm_cost_total(sh.cls, M, lane);
        lm_decide(r, cost_raw, prm, M, lane, (w == 0) ? trace : nullptr, dec MVO_STAMP_ARG);
        if (SPLIT && lane == 0) {  // the verdict, for the solver wavefront
          sh.ctl_radius = r.radius;
          sh.ctl_take = dec.take_z ? 1 : 0;
          sh.ctl_done = dec.done ? 1 : 0;
          sh.ctl_first = dec.first ? 1 : 0;
          sh.ctl_reuse = r.reuse_diag;
        }
        __syncthreads();  // B1c: the system at z and its gradient norm are complete
      } else {
        LmTables tb = lm_tables(sh, 0);
        if (prm.cls_ready) lm_prefix_suffix<false>(sh.cls, nullptr, tb.C1, tb.D, M, N, lane);
        else lm_tables_from_pairs(pairbuf, tb.C1, tb.D, M, N, lane);
        MVO_STAMP(3);
        lm_block_rows<0>(tb, br, lane, hsel ? sh.st.H : sh.Hz);
        lm_gradient_entry(tb, lane >= br.nB4 && lane < br.nB4 + 3 * M, br.k4, br.i4);
        wave0_sync();
        lm_decide(r, tb.D[(MVO_MAXM - 1) * MCX][9], prm, M, lane, (w == 0) ? trace : nullptr, dec MVO_STAMP_ARG);
      }
      MVO_STAMP(4);
      // ---------------- damping, next candidate ----------------
      double mcc_term = 0.0, pub_s = 0.0, pub_sp = 0.0;
      int go;
      if (SPLIT) go = lm_control_finish(r, sh, prm, lane, (w == 0) ? trace : nullptr, hsel, dec);
      else go = lm_step<LARGE>(r, sh, prm, M, n, lane, (w == 0) ? trace : nullptr, hsel, dec, mcc_term, pub_s,
                               pub_sp MVO_STAMP_ARG);
      if (lane == 0) sh.hsel = hsel;
      if (lane == 0) sh.go = go != 0;
      if (LARGE) __syncthreads();  // B2: the class wavefronts start on the point
      else wave0_sync();
      MVO_STAMP(5);
      if (!go) break;
      // off the critical path (k_lm<256>: the class wavefronts are at the pairs): the step's model cost change and
      // validity, and the half of the point only the product stage reads (behind B1 / this wavefront's own order)
      if (SPLIT) {
        mcc_term = lane < n ? sh.mcc_term[lane] : 0.0;
        go = sh.step_ok ? 1 : 2;
        if (go == 1) {  // the candidate the solver wavefront published
          const int k = lane < M ? lane : 0;
          r.cv[0] = sh.zvec[k][0]; r.cv[1] = sh.zvec[k][1]; r.cv[2] = sh.zvec[k][2];
          r.ct = sh.zt[k];
          pub_s = sh.zs[k];
          pub_sp = sh.zsp[k];
        }
      }
      lm_model_cost_change(r, mcc_term);
      if (go == 1) lm_publish_point_late(sh, M, lane, r.cv, r.ct, pub_s, pub_sp);
    }
    MVO_STAMP(2);
    if (prm.pass_mode && !prm.inline_eval) need_pass = true;
  }

  if (need_pass) {
    // Publish the unit baselines of the point to evaluate; k_pass runs next on the stream.  No global load
    // stands between a thread and its pairs: point and fixed-frame offsets come from LDS, (a, b) from the
    // loop counter.  A long pair (a, b >= M) has the baseline B_a + n_b v_b with B_a = s_a v_a + ... +
    // s_{M-1} v_{M-1} (left to right, IR.cpp:466-577) -- five vectors per point, formed once.  One reciprocal instead of three divisions: u differs from beta / |beta| by
    // <= 1 ulp, and pass and projection both read THIS u.
    __shared__ double sB[MVO_MAXM][3];
    if (tid < M) {
      double beta[3];
      free_baseline(sh, tid, M - 1, beta);
      sB[tid][0] = beta[0];
      sB[tid][1] = beta[1];
      sB[tid][2] = beta[2];
    }
    __syncthreads();
    double* pu = PERSIST ? su : pair_u + 4 * (size_t)win.pair_off;  // PERSIST: every workgroup keeps its own copy in LDS
    const int n_long = N - 1 > M ? N - 1 - M : 0;
    for (int q = tid; q < n_long * M; q += NT) {  // long pairs, frame-major: q = a * n_long + (b - M)
      const int a = q / n_long, b = M + (q - a * n_long);
      const int bb = b < FOFF_FRAMES ? b : 0;
      const double bx = sB[a][0] + sfoff[3 * bb], by = sB[a][1] + sfoff[3 * bb + 1], bz = sB[a][2] + sfoff[3 * bb + 2];
      const double nb = sqrt(bx * bx + by * by + bz * bz);
      const double inb = fast_rcp(nb);
      double* o = pu + 4 * (size_t)pair_index(N, a, b);
      o[0] = bx * inb;
      o[1] = by * inb;
      o[2] = bz * inb;
      o[3] = nb;
    }
    // the pairs (a, b < M): their own partial sums of the refined frames, no fixed frame; the last lanes take them
    const int ts = NT - 1 - tid, nshort = (M * (M + 1)) / 2;
    if (ts < nshort) {
      int a, b;
      upper_block(M, ts, a, b);
      double beta[3];
      free_baseline(sh, a, b, beta);
      const double nb = sqrt(beta[0] * beta[0] + beta[1] * beta[1] + beta[2] * beta[2]);
      const double inb = fast_rcp(nb);
      double* o = pu + 4 * (size_t)pair_index(N, a, b);
      o[0] = beta[0] * inb;
      o[1] = beta[1] * inb;
      o[2] = beta[2] * inb;
      o[3] = nb;
    }
  }
  if (!PERSIST || !need_pass) break;
  // ---- PERSIST: the pass at the candidate over this workgroup's tiles, then the grid-wide hand-over ----
  {
    __syncthreads();  // su is complete
    MVO_RSTAMP(0);  // the replicated evaluation and step, su
    const uint32_t bw0 = blockIdx.x * FOLD_WAVES;
    const uint32_t bw1 = bw0 + FOLD_WAVES < ra.n_waves ? bw0 + FOLD_WAVES : ra.n_waves;
    const uint32_t bt0 = s_wt[0], bt1 = s_wt[bw1 - bw0];
    if (ra.robust) pass_wave_tiles<true>(s_tiles, s_wt, bw0 + (uint32_t)wv, ra.n_waves, bt0, ra.cbuf, ra.cstride, nullptr, su, red_buf, lane, bt0, bw0);
    else pass_wave_tiles<false>(s_tiles, s_wt, bw0 + (uint32_t)wv, ra.n_waves, bt0, ra.cbuf, ra.cstride, nullptr, su, red_buf, lane, bt0, bw0);
    MVO_RSTAMP(1);  // pass over the own tiles
    block_store_partials(s_tiles, bt0, bt1, red_buf, ra.partial, true, bt0);
    MVO_RSTAMP(2);  // slots out (acknowledged)
    PassOut po;
    po.upairs = su;
    po.proj = ra.proj[par ^ 1];
    po.coherent_proj = 1;
    po.coherent_sums = 0;
    block_finish_pairs(ra.pair_ticket, s_tiles, bt0, bt1, ra.pairs_slots, red_buf, ra.partial, ra.pair_sums, po, bt0);
    MVO_RSTAMP(3);  // tickets, finished pairs added up and projected
    par ^= 1;
    proj = ra.proj[par];
    // every pair's projection is out once every workgroup has arrived (each waits for its own stores first)
    if (!grid_sync(ra.gs, ra.n_wg, grid_epoch) || turn > prm.max_it + 8) {
      grid_fail = true;
      break;
    }
    MVO_RSTAMP(4);  // grid barrier
    need_pass = false;
  }
  }  // for (turn)
  if (PERSIST && (grid_fail || !lead_wg)) return;  // only workgroup 0 reports; a failed wait reports nothing (the host repeats the solve)
  // the solver state leaves the registers: LDS image, then HBM (pass mode re-enters with it)
  constexpr bool ONE_LAUNCH = (SPEC >= 1 && SPEC <= 4) || SPEC == 7;  // the solve ends in this launch: nobody reads the state again
  if (wave0) {
    regs_store(r, sh.st, lane);
    if (lane == 0) sh.phase = hsel;
  }
  if (ONE_LAUNCH) {
    if (!wave0) return;
    wave0_sync();  // wavefront 0 reads back its own image below
  } else {
    __syncthreads();
    if (sh.phase) {  // bring the accepted system home before the state leaves LDS
      for (int i = tid; i < MVO_MAXC * MVO_MAXC; i += NT) sh.st.H[i] = sh.Hz[i];
      __syncthreads();
    }
    for (int i = tid; i < (int)(sizeof(LmState) / 8); i += NT)
      reinterpret_cast<double*>(gst)[i] = reinterpret_cast<const double*>(&sh.st)[i];
  }
#ifdef MVO_LM_STAMPS
  if (w == 0 && tid == 0) stamp_lds_[7] += __builtin_amdgcn_s_memtime() - stamp_prev_;  // slot 7: last stamp -> state out
  __syncthreads();
  if (w == 0 && tid < 16) mvo_lm_stamps[tid] += stamp_lds_[tid];
#endif
  if (wave0 && sh.st.status != 0) {
    const LmState& st = sh.st;
    LmResult* __restrict__ res = results + w;
    if (lane < MVO_MAXM) {
      // Ceres hands the lowest-cost iterate back only when the summary is usable (CONVERGENCE /
      // NO_CONVERGENCE: Solver::Summary::IsSolutionUsable); after FAILURE the user's arrays keep
      // their INITIAL values, and IR.cpp:240-246 then writes s(reverseScale(scale_0)) and vec_0.
      const bool usable = st.termination != 2;
      const double* __restrict__ x0 = init_x + ((size_t)w * MVO_MAXM + lane) * 4;
      const double bt = usable ? st.b_t[lane] : x0[3];
      res->vec[lane][0] = usable ? st.b_vec[lane][0] : x0[0];
      res->vec[lane][1] = usable ? st.b_vec[lane][1] : x0[1];
      res->vec[lane][2] = usable ? st.b_vec[lane][2] : x0[2];
      res->scale[lane] = d_scale(bt, prm.max_len, prm.min_len);
    }
    if (lane == 0) {
      res->initial_cost = st.initial_cost;
      res->final_cost = st.final_cost;
      res->termination = st.termination;
      res->reason = st.reason;
      res->iterations = st.ntrace - 1;
      res->nsucc = st.nsucc;
      res->nunsucc = st.nunsucc;
    }
    // the record may live in pinned host memory, and the host may be polling its mark instead of waiting
    // for the stream: every field is out (system scope) before the mark is
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) {
      __hip_atomic_store(&res->pad, kResultWritten, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      win_status[w] = 1;
      atomicAdd(done_count, 1);
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_class_sums (observation-partitioned pass mode): this rank's class totals from its pairs' projected
// moments -- what k_lm's per-pair stage leaves in LmShared::cls / kk -- as MVO_CLS_DOUBLES per window in
// HBM.  The projection is linear in the pair's sums and the class totals are sums of projections, so the
// ranks' class totals add up to the window's: ONE all-reduce of 350 fp64 per window and evaluation
// ("the already-contracted" form) instead of 8 per pair, and no rank repeats the per-pair
// stage.  One workgroup per window, one wavefront per refined frame a: its long class (a, b >= M), lanes
// over b (up to four rounds, loads requested together); the short pairs (a, b < M) are classes of their
// own and are copied.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * MVO_MAXM) void k_class_sums(int M, int N, int P, const double* __restrict__ proj,
                                                             double* __restrict__ cls_out) {
  constexpr int ROUNDS = 4;
  const int w = blockIdx.x, lane = threadIdx.x & 63;
  const int a = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double* __restrict__ pw = proj + 10 * (size_t)w * P;
  double* __restrict__ out = cls_out + (size_t)w * MVO_CLS_DOUBLES;
  // entries no class owns (a > j, a >= M) are never read by k_lm; they are zeroed because the all-reduce adds them
  for (int i = threadIdx.x; i < MVO_CLS_DOUBLES; i += 64 * MVO_MAXM) {
    const int row = i / 10, ra = row / MCX, rj = row - ra * MCX;
    const bool owned = row < MVO_MAXM * MCX ? (ra < M && rj >= ra && rj <= M) : (row - MVO_MAXM * MCX < M);
    if (!owned) out[i] = 0.0;
  }
  if (a >= M) return;
  const int n_rounds = ((N - 1 > M ? N - 1 - M : 0) + 63) >> 6;
  double o[ROUNDS][10];
  bool ok[ROUNDS];
#pragma unroll
  for (int rr = 0; rr < ROUNDS; ++rr) {
    const int b = M + lane + 64 * rr;
    ok[rr] = b <= N - 2;
    if (rr < n_rounds) proj_load(pw, (size_t)(ok[rr] ? pair_index(N, a, b) : 0), o[rr]);
  }
  double acc[10];
#pragma unroll
  for (int v = 0; v < 10; ++v) {
    acc[v] = 0.0;
#pragma unroll
    for (int rr = 0; rr < ROUNDS; ++rr)
      if (rr < n_rounds) acc[v] += ok[rr] ? o[rr][v] : 0.0;
  }
  double t8, t2;
  wave_reduce10(acc, lane, t8, t2);
  double* __restrict__ cm = out + (a * MCX + M) * 10;
  if ((lane & 7) == 0) cm[reduce8_value_of_lane(lane)] = t8;
  if ((lane & 31) == 0) cm[8 + (lane >> 5)] = t2;
  // short pairs (a, j), a <= j < M: lane = (j - a) * 10 + v
  const int jj = a + lane / 10, v = lane % 10;
  if (jj < M && lane < 10 * MVO_MAXM) {
    const double x = pw[10 * (size_t)pair_index(N, a, jj) + v];
    out[(a * MCX + jj) * 10 + v] = x;
    if (jj == a) out[(MVO_MAXM * MCX + a) * 10 + v] = x;  // kk[a]
  }
}

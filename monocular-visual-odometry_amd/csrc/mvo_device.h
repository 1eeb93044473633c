// mvo_device.h -- device-side records and scalar math shared by the gfx950 kernels.
//
// Reference = /root/reference/mvo/src/algorithms/IterativeRefinement.cpp ("IR.cpp").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#define MVO_MAXM 5
#define MVO_MAXC 15
#define MVO_FRAME_DOUBLES 13  // R(9) vec(3) scale(1): mvo_hip_frame

// ---- tables built on the host at upload -------------------------------------------------
struct WinDev {
  uint32_t frame_off;  // first frame record of the window
  uint32_t pair_off;   // first PairDev of the window
  uint32_t N, M, P;
  uint32_t pad[3];
};

struct PairDev {
  uint32_t slot_begin, slot_end;  // partial-sum slots owned by this pair
  uint32_t a, b;                  // frame indices within the window (IR.cpp:116,134)
  uint32_t src_start, count;      // the pair's residual blocks in the caller-ordered ray arrays
};

// One wave-tile: a run of residual blocks of ONE frame pair, processed by one wavefront.
struct TileDev {
  uint32_t pair;       // global pair index (into PairDev / pair_u)
  uint32_t win;        // window index
  uint32_t src_start;  // first residual block in the caller-ordered AoS ray arrays
  uint32_t c_start;    // first element in the padded SoA planes of c (multiple of 2)
  uint32_t count;      // residual blocks in this tile
  uint32_t slot;       // partial-sum slot this tile adds into (never spans workgroups)
  uint32_t f0, f1;     // global frame records of R0 = frames[a].R and R1 = frames[b+1].R
};

struct SolveParams {
  double ftol, gtol, ptol;
  double min_len, max_len;
  int32_t max_it;
  int32_t use_loss;
  int32_t fix_length;
  int32_t pass_mode;   // 1: every evaluation is an observation pass (k_pass)
  int32_t trace_cap;
  int32_t inline_eval;  // 1: k_lm evaluates the observations itself (small / many windows)
  int32_t fresh;        // 1: first launch of a solve -- the stored state is stale, start from zeros
  int32_t win_base;     // k_lm: window of workgroup 0 (launches over a range of the windows)
  int32_t M, N, P;      // every window of an upload has the same shape
  int32_t x0_ready;     // pass mode: the sums of the evaluation at x0 exist already (fused fold)
  int32_t proj_ready;   // pass mode, passes as launches of their own: their tails left every pair's projected moments
                        // in `proj` (project_pair8, mvo_kernels.h)
  int32_t cls_ready;    // ... observation-partitioned: the class totals of the evaluation, summed over the ranks, are
                        // in `cls_in` (k_class_sums + all-reduce)
  // One window of at most MVO_SMALL_PAIRS frame pairs that evaluates its own observations (the reference's shipped
  // window: three): the pairs' offsets into the ray arrays travel with the kernel arguments, so the first loads of the
  // launch are the rays themselves and not a table whose entries say where they are (one cold round trip, ~0.8 us of
  // a 26-us call).  small_tab = 0: the pair table in memory.
  int32_t small_tab;
  uint32_t small_offs[9];  // pair p: residual blocks small_offs[p] .. small_offs[p + 1]
};
#define MVO_SMALL_PAIRS 8
#define MVO_CLS_DOUBLES 352  // per window: cls[MAXM * (MAXM + 1)][10], kk[MAXM][10], padded to a multiple of 4

struct TraceRec {  // == mvo_hip_iteration
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int32_t step_is_valid, step_is_successful;
};

enum { PH_START = 0, PH_EVAL_X0 = 1, PH_EVAL_CAND = 2 };

// Per-window Levenberg-Marquardt state; lives in HBM between launches (pass mode) and in LDS
// while k_lm runs.  sizeof is a multiple of 8.
struct LmState {
  double x_vec[MVO_MAXM][3], x_t[MVO_MAXM];  // accepted point (global parameters)
  double c_vec[MVO_MAXM][3], c_t[MVO_MAXM];  // candidate
  double b_vec[MVO_MAXM][3], b_t[MVO_MAXM];  // lowest-cost iterate = what Ceres hands back
  double H[MVO_MAXC * MVO_MAXC];             // J^T J at x (active columns, row stride MAXC)
  double g[MVO_MAXC];                        // J^T r at x
  double jscale[MVO_MAXC], diag[MVO_MAXC], delta[MVO_MAXC];
  double x_cost, x_norm, min_cost, model_cost_change;
  double radius, decrease_factor, grad_max_norm;
  double initial_cost, final_cost;
  double se_min, se_cur, se_ref, se_cand, se_acc_ref, se_acc_cand;  // TrustRegionStepEvaluator
  int32_t se_nonmono, reuse_diagonal;
  int32_t iteration, num_invalid;
  int32_t nsucc, nunsucc;
  int32_t status;  // 0 running, 1 terminated
  int32_t termination, reason;
  int32_t phase;
  int32_t ntrace;
  int32_t pending_invalid;  // the step that led to the point under evaluation turned out invalid: the evaluation is
                            // ignored and HandleInvalidStep runs instead of the verdict (lm_decide)
};

struct LmResult {  // (200 B: what crosses PCIe per window and what the host's write-back loop reads -- nothing it does not use)
  double vec[MVO_MAXM][3];
  double scale[MVO_MAXM];  // s(t): the write-back value of IR.cpp:240-246, so that a download is a copy
  double initial_cost, final_cost;
  int32_t termination, reason, iterations, nsucc, nunsucc, pad;  // pad = kResultWritten once written
};
constexpr int32_t kResultWritten = 0x600D;

// ---- scalar math ----------------------------------------------------------------------

// IR.cpp:298-309 scaleTemplated<double>
__device__ __forceinline__ double d_scale(double t, double max_len, double min_len) {
  const double e = exp(-1.0 * t);
  if (isinf(e)) return min_len;
  return min_len + ((max_len - min_len) / (1.0 + e));
}

// derivative as Ceres' Jet arithmetic yields it (0 in the Inf branch)
__device__ __forceinline__ double d_dscale(double t, double max_len, double min_len) {
  const double e = exp(-1.0 * t);
  if (isinf(e)) return 0.0;
  const double g = 1.0 + e;
  return (-(max_len - min_len) / (g * g)) * (-e);
}

// 1/x for x in a sane range (pivots, 1 + a^2): v_rcp_f64 seed + two Newton steps (<= 1 ulp),
// a third of the instructions of the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// 1/sqrt(x) for x in a sane range (squared baseline lengths): v_rsq_f64 seed + two coupled Newton steps
// (Goldschmidt: g -> sqrt(x), h -> 1 / (2 sqrt(x))), <= 1 ulp; nine instructions where sqrt + reciprocal take ~35
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  h = fma(h, e, h);
  return h + h;
}

// exp(x) in ~20 instructions for k_lm's critical path (the library routine inlines to 106): x = k ln2 + r,
// |r| <= ln2 / 2, Taylor polynomial of degree 13 (remainder 4e-18 relative), scaled by v_ldexp_f64, which
// saturates to +inf / flushes towards 0 like the library's result does.  The argument is clamped to
// +-1100 first so that the integer conversion never saturates (exp(+-1100) is already inf / 0).
// Measured against libm over [-745, 710]: scripts/exp/rcp_precision.hip.
__device__ __forceinline__ double fast_exp(double x) {
  const double xc = fmin(fmax(x, -1100.0), 1100.0);
  const double kf = rint(xc * 1.4426950408889634074);
  double r = fma(-kf, 6.93147180369123816490e-01, xc);  // ln2, high part (low 21 bits zero: kf * hi is exact)
  r = fma(-kf, 1.90821492927058770002e-10, r);           // ln2, low part
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// IR.cpp:282-296 + :334-346: x (+) (a,b) = (B*A)*x
__device__ __forceinline__ void d_plus(const double x[3], double a, double b, double out[3]) {
  const double ida = fast_rcp(1.0 + a * a), idb = fast_rcp(1.0 + b * b);
  const double ca = (1.0 - a * a) * ida, sa = (2.0 * a) * ida, msa = (-2.0 * a) * ida;
  const double cb = (1.0 - b * b) * idb, sb = (2.0 * b) * idb, msb = (-2.0 * b) * idb;
  // A = [ca msa 0; sa ca 0; 0 0 1], B = [cb 0 sb; 0 1 0; msb 0 cb]; BA = B*A
  const double BA[9] = {cb * ca, cb * msa, sb,
                        sa, ca, 0.0,
                        msb * ca, msb * msa, cb};
  out[0] = BA[0] * x[0] + BA[1] * x[1] + BA[2] * x[2];
  out[1] = BA[3] * x[0] + BA[4] * x[1] + BA[5] * x[2];
  out[2] = BA[6] * x[0] + BA[7] * x[1] + BA[8] * x[2];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Summed baseline of pair (a,b) at the point (zvec, zs): IR.cpp:137-144 (offset quirk: only
// frame b itself is added when b >= M) and IR.cpp:466-577 (s_a v_a + ... left to right),
// then IR.cpp:254-255 (add offset, normalise).  fr = the window's frame records.
__device__ __forceinline__ void pair_unit_baseline(const double* __restrict__ fr, int M, int a, int b,
                                                   const double (*zvec)[3], const double* zs,
                                                   double u[3], double& nb) {
  const int kmax = b < M - 1 ? b : M - 1;
  double bx = 0.0, by = 0.0, bz = 0.0;
  for (int k = a; k <= kmax; ++k) {
    const double tx = zs[k] * zvec[k][0], ty = zs[k] * zvec[k][1], tz = zs[k] * zvec[k][2];
    if (k == a) { bx = tx; by = ty; bz = tz; } else { bx += tx; by += ty; bz += tz; }
  }
  if (b >= M) {
    const double* f = fr + MVO_FRAME_DOUBLES * b;
    bx += f[12] * f[9];
    by += f[12] * f[10];
    bz += f[12] * f[11];
  }
  nb = sqrt(bx * bx + by * by + bz * bz);
  u[0] = bx / nb;
  u[1] = by / nb;
  u[2] = bz / nb;
}

/* mvo_ba_oracle.h -- CPU oracle of the explicit-landmark window bundle adjustment (SURVEY.md 8f,
 * rank 1).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it; nothing under monocular-visual-odometry_amd/ does.
 *
 * PARITY UNPINNED: the reference (FranekStark/monocular-visual-odometry) has NO explicit-landmark
 * bundle adjustment -- its refinement (mvo/src/algorithms/IterativeRefinement.cpp) is the
 * landmark-free epipolar formulation that oracle/mvo_oracle.c restates.  This problem exists only
 * because BASELINE.json's north_star asks for pose-landmark blocks, a Schur complement and a
 * reduced camera system; there is no reference file, test or fixture to pin it to.  The oracle
 * below is therefore a self-oracle: a plain, independent-in-structure (dense, scalar, textbook)
 * CPU statement of the same mathematical problem, checked by finite differences and by
 * recovering known synthetic scenes (tests/test_ba_oracle.py).
 *
 * Problem.  N keyframes with camera-to-world rotation R_i (row-major) and position p_i, the same
 * convention as RefinementFrame.R of the reference (bearing = R_i^T (X - p_i)); L landmarks X_j;
 * observations = unit bearing rays m of landmark j in frame i, grouped by landmark (CSR).
 * Residual (2 per observation): r = (q_x/q_z - m_x/m_z, q_y/q_z - m_y/m_z), q = R_i^T (X_j - p_i).
 * Unknowns: the newest M frames (indices 0..M-1, as in the reference's window: newest first),
 * 6 each -- local rotation omega (R <- R Exp(omega)) and position -- and all landmarks, 3 each.
 * Frames M..N-1 are fixed (they anchor the gauge; N - M >= 2 fixes the scale as well).
 * Loss: none, Cauchy or Huber on s = |r|^2 with scale a (rho'' <= 0: plain IRLS, weight
 * sqrt(rho') on residual and Jacobian rows).  cost = 1/2 sum rho(s).
 * Solver: Levenberg-Marquardt, Ceres-style trust region (initial radius 1e4, damping
 * D^2 = clamp(diag(J^T J), 1e-6, 1e32) / radius, step quality rho > 1e-3 accepts,
 * radius /= max(1/3, 1 - (2 rho - 1)^3) on accept, radius /= 2, 4, 8 ... on consecutive rejects,
 * monotonic), linear solve by Schur complement on the landmarks + dense Cholesky.
 */
#ifndef MVO_BA_ORACLE_H
#define MVO_BA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double R[9]; /* camera-to-world, row-major */
  double p[3];
} mvo_ba_pose;

typedef struct {
  int32_t max_num_iterations;
  int32_t loss; /* 0 none, 1 Cauchy, 2 Huber */
  double loss_scale;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius; /* <= 0: 1e4 */
} mvo_ba_options;

typedef struct {
  int32_t termination; /* 0 convergence, 1 no convergence (iteration limit), 2 failure */
  int32_t iterations, num_successful_steps, num_unsuccessful_steps;
  double initial_cost, final_cost;
} mvo_ba_summary;

typedef struct {
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int32_t step_is_valid, step_is_successful;
} mvo_ba_iteration;

/* residuals (2 per observation, unweighted) and cost (with the loss) at the given point */
void mvo_ba_oracle_residuals(uint32_t N, const mvo_ba_pose *poses, uint32_t L, const double *X,
                             const uint64_t *lm_offsets, const uint32_t *obs_frame, const double *obs_ray,
                             const mvo_ba_options *opt, double *residuals, double *cost);

/* Reduced camera system at the given point with damping at `radius`:
 * S (6M x 6M, row-major, symmetric) and rhs (6M) of  S dc = -rhs,
 * S = B + Dc^2 - sum_j E_j (C_j + Dl_j^2)^-1 E_j^T,  rhs = g_c - sum_j E_j (C_j + Dl_j^2)^-1 g_lj.
 * gradient (6M + 3L, cameras first) optional. */
void mvo_ba_oracle_reduced(uint32_t N, uint32_t M, const mvo_ba_pose *poses, uint32_t L, const double *X,
                           const uint64_t *lm_offsets, const uint32_t *obs_frame, const double *obs_ray,
                           const mvo_ba_options *opt, double radius, double *S, double *rhs, double *gradient,
                           double *cost);

/* Analytic Jacobian rows of one observation: jc (2x6: omega, p), jx (2x3) -- for the finite-difference test */
void mvo_ba_oracle_obs_jacobian(const mvo_ba_pose *pose, const double *X, const double *ray, double r[2],
                                double jc[12], double jx[6]);

/* R <- R Exp(omega), p <- p + dp */
void mvo_ba_oracle_pose_plus(const mvo_ba_pose *in, const double delta[6], mvo_ba_pose *out);

/* full solve; poses[0..M) and X are updated in place; trace may be NULL */
int mvo_ba_oracle_refine(uint32_t N, uint32_t M, mvo_ba_pose *poses, uint32_t L, double *X,
                         const uint64_t *lm_offsets, const uint32_t *obs_frame, const double *obs_ray,
                         const mvo_ba_options *opt, mvo_ba_summary *summary, mvo_ba_iteration *trace,
                         int32_t trace_capacity, int32_t *trace_rows);

#ifdef __cplusplus
}
#endif
#endif

/*
 * mvo_oracle.h -- CPU restatement (fp64, plain C) of the reference's sliding-window
 * refinement solver.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is linked into,
 * imported by, or executed from the product library (libmvo_hip).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * What it restates (all paths relative to /root/reference/mvo/src):
 *   algorithms/IterativeRefinement.cpp:15-250   refine(): parameter setup, pair enumeration,
 *                                               offset quirk, write-back
 *   algorithms/IterativeRefinement.cpp:252-263  CostFunctionBase::cost  (epipolar residual)
 *   algorithms/IterativeRefinement.cpp:282-296  baseLineTemplated       (sphere retraction)
 *   algorithms/IterativeRefinement.cpp:298-309  scaleTemplated          (sigmoid length)
 *   algorithms/IterativeRefinement.cpp:348-368  reverseScale            (inverse sigmoid)
 *   algorithms/IterativeRefinement.cpp:466-577  CostFunction1..5        (summed baselines)
 *   operations/FeatureOperations.cpp:117-122    normFeatures
 *
 * The solver arithmetic itself (autodiff, Levenberg-Marquardt trust region, dense QR,
 * loss corrector) lives in Ceres Solver, a third-party dependency that is NOT vendored in
 * the reference and NOT pinned to a version (mvo/CMakeLists.txt:44, find_package(Ceres
 * REQUIRED); C++11 + ROS melodic imply Ceres 1.13/1.14).  Ceres, Eigen, OpenCV and ROS are
 * all absent from this image, so the reference cannot be built here.  The LM loop below
 * restates Ceres 1.14's published algorithm (TrustRegionMinimizer, LevenbergMarquardtStrategy,
 * TrustRegionStepEvaluator, DenseQRSolver, Corrector, CauchyLoss) from its documentation.
 *
 * PARITY STATUS
 *   residual + Jacobian : pinned BY FORMULA to the reference source (Ceres autodiff is the
 *                         exact derivative; the oracle carries a dual-number evaluation of the
 *                         reference's own expression tree to prove its analytic Jacobian).
 *   LM trajectory/result: PARITY UNPINNED -- the reference ships no tests, golden vectors or
 *                         fixtures for this path, and Ceres is absent.
 *                         The converged minimiser is cross-checked against an independent
 *                         scipy.optimize.least_squares run (tests/golden/make_golden.py).
 */
#ifndef MVO_ORACLE_H
#define MVO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* IterativeRefinement::RefinementFrame (IterativeRefinement.hpp:171-175): R row-major. */
typedef struct {
  double R[9];
  double vec[3];
  double scale;
} mvo_oracle_frame;

/* Arguments of IterativeRefinement::refine (IterativeRefinement.hpp:181-192) + fixLength
 * (read through nowFrame.getParameters(), IterativeRefinement.cpp:105). */
typedef struct {
  int32_t max_num_iterations;
  int32_t num_threads; /* ignored by the oracle: it is a scalar port */
  double function_tolerance;
  double gradient_tolerance;
  double parameter_tolerance;
  int32_t use_loss_function; /* CauchyLoss(0.5), IterativeRefinement.cpp:85-88 */
  int32_t fix_length;
  double lowest_length;
  double highest_length;
} mvo_oracle_options;

enum {
  MVO_ORACLE_CONVERGENCE = 0,
  MVO_ORACLE_NO_CONVERGENCE = 1,
  MVO_ORACLE_FAILURE = 2
};

enum {
  MVO_ORACLE_REASON_NONE = 0,
  MVO_ORACLE_REASON_MAX_ITERATIONS = 1,
  MVO_ORACLE_REASON_GRADIENT_TOLERANCE = 2,
  MVO_ORACLE_REASON_PARAMETER_TOLERANCE = 3,
  MVO_ORACLE_REASON_FUNCTION_TOLERANCE = 4,
  MVO_ORACLE_REASON_MIN_RADIUS = 5,
  MVO_ORACLE_REASON_INVALID_STEPS = 6,
  MVO_ORACLE_REASON_LINEAR_SOLVER = 7,
  MVO_ORACLE_REASON_EVALUATION_FAILED = 8 /* non-finite residuals at the initial point */
};

typedef struct {
  int32_t termination;
  int32_t reason;
  int32_t iterations; /* LM steps attempted (Ceres' last IterationSummary::iteration) */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t reserved;
  double initial_cost;
  double final_cost;
} mvo_oracle_summary;

/* One row per Ceres IterationSummary (iteration 0 = the starting point). */
typedef struct {
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  int32_t step_is_valid;
  int32_t step_is_successful;
} mvo_oracle_iteration;

/* Threads used by mvo_oracle_refine / mvo_oracle_eval (default 1 = the serial parity path; > 1 is for
 * bench.py's cpu_baseline timing legs only: sums are formed per pair and per row range, so results
 * differ from the serial ones in the last bits). */
void mvo_oracle_set_threads(int32_t n);
int32_t mvo_oracle_get_threads(void);

/* |P| = M(N-1) - M(M-1)/2 and the idx-th pair (a,b) in the reference's loop order
 * (IterativeRefinement.cpp:116,134). */
int32_t mvo_oracle_num_pairs(uint32_t M, uint32_t N);
void mvo_oracle_pair(uint32_t M, uint32_t N, uint32_t idx, uint32_t *a, uint32_t *b);

double mvo_oracle_reverse_scale(double length, double max_len, double min_len);
double mvo_oracle_scale(double t, double max_len, double min_len);
double mvo_oracle_scale_derivative(double t, double max_len, double min_len);
void mvo_oracle_plus(const double x[3], const double delta[2], double x_plus_delta[3]);
void mvo_oracle_norm_features(double *rays, uint64_t n); /* in place, 3 doubles per ray */

/*
 * Residuals and local Jacobian at the INITIAL point (t_k = reverseScale(scale_k)).
 *   pair_offsets[n_pairs+1] : CSR over residual blocks, pairs in reference order
 *   m0, m1                  : unit rays, 3 doubles per residual block (m0 in frame a, m1 in
 *                             frame b+1)
 *   residuals[n_obs]        : raw r (no loss correction)
 *   jacobian[n_obs*3M]      : row-major, columns per refined frame k = (da, db, dt); raw
 * Returns 0 on success.
 */
int32_t mvo_oracle_eval(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                        const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                        const double *m0, const double *m1, double *residuals,
                        double *jacobian, double *cost);

/* The same at an explicit point: point[4k..4k+3] = (vec_k, t_k); NULL = the initial point. */
int32_t mvo_oracle_eval_at(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                           const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                           const double *m0, const double *m1, const double *point, double *residuals,
                           double *jacobian, double *cost);

/* Same outputs, Jacobian obtained by forward-mode dual numbers pushed through the
 * reference's own expression tree (what Ceres' AutoDiffCostFunction +
 * AutoDiffLocalParameterization compute). */
int32_t mvo_oracle_eval_autodiff(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                                 const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                                 const double *m0, const double *m1, double *residuals,
                                 double *jacobian, double *cost);

/* Per-pair second moments S_p = sum c c^T (xx,xy,xz,yy,yz,zz), c = (R_a m0) x (R_{b+1} m1). */
int32_t mvo_oracle_pair_moments(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                                const uint64_t *pair_offsets, const double *m0,
                                const double *m1, double *moments /* n_pairs*6 */);

/*
 * The whole refine(): frames[0..M) get vec and scale overwritten (even after FAILURE,
 * IterativeRefinement.cpp:235-246).  trace may be NULL; at most trace_capacity rows written,
 * *trace_len receives the number of iteration summaries produced.
 */
int32_t mvo_oracle_refine(mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                          const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                          const double *m0, const double *m1, mvo_oracle_summary *summary,
                          mvo_oracle_iteration *trace, int32_t trace_capacity,
                          int32_t *trace_len);

/* The same from an explicit starting point: point[4k..4k+3] = (vec_k, t_k); NULL = mvo_oracle_refine. */
int32_t mvo_oracle_refine_from(mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                               const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                               const double *m0, const double *m1, const double *point,
                               mvo_oracle_summary *summary, mvo_oracle_iteration *trace,
                               int32_t trace_capacity, int32_t *trace_len);

#ifdef __cplusplus
}
#endif
#endif /* MVO_ORACLE_H */

/* mvo_hip_ba.h -- C ABI of libmvo_hip's explicit-landmark window bundle adjustment.
 *
 * NO COUNTERPART IN THE REFERENCE.  FranekStark/monocular-visual-odometry refines baselines with a
 * landmark-free epipolar cost (mvo/src/algorithms/IterativeRefinement.cpp:15-250, served by
 * mvo_hip.h); it never triangulates or adjusts structure.  This entry-point family exists because
 * BASELINE.json's north_star asks for pose-landmark / landmark-landmark blocks, landmark-block
 * inversion, a Schur complement into the reduced camera system and a batched dense Cholesky
 * (SURVEY.md section 8f, rank 1).  Parity is therefore "unpinned": results are checked against the
 * self-oracle oracle/mvo_ba_oracle.c only.
 *
 * Conventions follow the reference where it has any: frame 0 is the newest keyframe
 * (Refiner.cpp:62-68), R is camera-to-world, row-major, exactly RefinementFrame.R
 * (IterativeRefinement.hpp:171-175), observations are unit bearing rays as normFeatures returns
 * them (FeatureOperations.cpp:117-122).  The newest M frames (6 DoF each: R <- R Exp(omega),
 * p <- p + dp) and all landmarks (3 DoF) are refined; frames M..N-1 stay fixed.
 *
 * Problem: residual (2 per observation) r = (q_x/q_z - m_x/m_z, q_y/q_z - m_y/m_z),
 * q = R_i^T (X_j - p_i); cost = 1/2 sum rho(|r|^2), rho = identity, Cauchy b log(1 + s/b) or Huber
 * (b = loss_scale^2; rho'' <= 0, so the corrector is plain IRLS: sqrt(rho') on residual and rows).
 * Solver rules (Ceres-style trust-region Levenberg-Marquardt, monotonic):
 *   per iteration: linearise at x; stop (convergence) if max |gradient| <= gradient_tolerance;
 *   damping D^2 = clamp(diag(J^T J), 1e-6, 1e32) / radius on cameras and landmarks;
 *   landmarks eliminated: S = B + Dc^2 - sum_j E_j (C_j + Dl_j^2)^-1 E_j^T, dense Cholesky of S;
 *   a failed factorisation, a non-finite step or model_cost_change <= 0 is an invalid step
 *   (radius /= 2; five in a row = failure);
 *   stop (convergence) if |step| <= parameter_tolerance (|x| + parameter_tolerance), |x| over the
 *   free positions and the landmarks;
 *   rho = cost_change / model_cost_change; rho > 1e-3 accepts (radius /= max(1/3, 1 - (2 rho - 1)^3),
 *   capped at 1e16), otherwise radius /= 2, 4, 8 ... on consecutive rejections;
 *   stop (convergence) after an accepted step with |cost_change| <= function_tolerance * cost,
 *   or when radius <= 1e-32; stop (no convergence) at max_num_iterations.
 */
#ifndef MVO_HIP_BA_H
#define MVO_HIP_BA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MVO_HIP_BA_MAX_FREE_FRAMES 28 /* reduced camera system <= 168 x 168: lives in one CU's LDS */
#define MVO_HIP_BA_MAX_FRAMES 65535

#define MVO_HIP_BA_LOSS_NONE 0
#define MVO_HIP_BA_LOSS_CAUCHY 1
#define MVO_HIP_BA_LOSS_HUBER 2

typedef struct mvo_hip_ba_context *mvo_hip_ba_handle;

typedef struct {
  double R[9]; /* camera-to-world, row-major */
  double p[3];
} mvo_hip_ba_pose; /* 96 bytes */

typedef struct {
  int32_t max_num_iterations;
  int32_t loss; /* MVO_HIP_BA_LOSS_* */
  double loss_scale;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius; /* <= 0: 1e4 */
} mvo_hip_ba_options; /* 48 bytes */

/* observations grouped by landmark (CSR): landmark j owns [landmark_offsets[j], landmark_offsets[j+1]) */
typedef struct {
  uint32_t n_landmarks;
  uint32_t reserved;
  const uint64_t *landmark_offsets; /* n_landmarks + 1 */
  const uint32_t *frame;            /* frame index of every observation */
  const double *rays;               /* 3 per observation, camera frame, z > 0 */
} mvo_hip_ba_observations;

typedef struct {
  int32_t termination; /* 0 convergence, 1 iteration limit, 2 failure */
  int32_t iterations, num_successful_steps, num_unsuccessful_steps;
  double initial_cost, final_cost;
} mvo_hip_ba_summary;

typedef struct {
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int32_t step_is_valid, step_is_successful;
} mvo_hip_ba_iteration;

typedef struct {
  double upload_seconds, solve_seconds, device_seconds;
  double schur_seconds;   /* sum over the k_ba_schur launches of the last solve */
  int32_t schur_launches; /* = linearisations */
  int32_t reserved;
  uint64_t observations;
  uint64_t schur_bytes;   /* algorithmic bytes of one k_ba_schur launch */
  uint64_t schur_flops;   /* algorithmic fp64 flops of one k_ba_schur launch: 96 per observation,
                           * 380 per observation in a free frame, 216 per pair of those within a landmark */
} mvo_hip_ba_timing;

/* status codes are those of mvo_hip.h (0 = ok) */
int mvo_hip_ba_create(int device_ordinal, mvo_hip_ba_handle *out);
void mvo_hip_ba_destroy(mvo_hip_ba_handle h);

/* one-call form: poses[0..M) and landmarks (3 per landmark) are refined in place */
int mvo_hip_ba_refine(mvo_hip_ba_handle h, mvo_hip_ba_pose *poses, uint32_t M, uint32_t N, double *landmarks,
                      const mvo_hip_ba_observations *observations, const mvo_hip_ba_options *options,
                      mvo_hip_ba_summary *summary);

/* split form: observations stay resident in HBM between solves; solve always restarts from the
 * uploaded point */
int mvo_hip_ba_upload(mvo_hip_ba_handle h, const mvo_hip_ba_pose *poses, uint32_t M, uint32_t N,
                      const double *landmarks, const mvo_hip_ba_observations *observations,
                      const mvo_hip_ba_options *options);
int mvo_hip_ba_solve(mvo_hip_ba_handle h);
int mvo_hip_ba_download(mvo_hip_ba_handle h, mvo_hip_ba_pose *poses, double *landmarks, mvo_hip_ba_summary *summary);

/* parity/debug, at the uploaded point: residuals (2 per observation, caller's order; may be NULL)
 * and the cost with the loss */
int mvo_hip_ba_eval(mvo_hip_ba_handle h, double *residuals, double *cost);
/* parity/debug, at the uploaded point: the damped reduced camera system S (6M x 6M, row-major),
 * rhs (6M), raw camera gradient (6M) and the largest gradient entry over cameras and landmarks */
int mvo_hip_ba_reduced(mvo_hip_ba_handle h, double radius, double *S, double *rhs, double *camera_gradient,
                       double *gradient_max_norm, double *cost);

/* Landmark-partitioned solve over several GPUs (BASELINE.json configs[3]: "partitioned over 8 GPUs
 * with RCCL all-reduce of Schur blocks").  Every rank uploads ALL poses and ITS range of the
 * landmarks with their observations.  The reduced camera system and the step sums are additive
 * over landmarks; the caller all-reduces them between the calls and every rank then takes the
 * identical LM step (no broadcast).  One iteration:
 *     partial_linearize;  exchange(0, out) -> all-reduce SUM -> exchange(0, in);
 *                         exchange(1, out) -> all-reduce MAX -> exchange(1, in);
 *     partial_solve;      exchange(2, out) -> all-reduce SUM -> exchange(2, in);
 *     partial_decide(1, &done)
 * preceded once by partial_begin; exchange(2 ...); partial_decide(0, &done).  When done is set the
 * results are ready for mvo_hip_ba_download (poses identical on every rank, landmarks = the rank's
 * own).  exchange copies between the library's buffers and a caller-owned DEVICE buffer
 * (direction 0: library -> buffer, 1: buffer -> library); sizes: which 0:
 * 18 M (M + 1) + 18 M + 1 doubles, which 1: 1, which 2: 4.  Every call drains the stream. */
int mvo_hip_ba_partial_begin(mvo_hip_ba_handle h);
int mvo_hip_ba_partial_linearize(mvo_hip_ba_handle h);
int mvo_hip_ba_partial_solve(mvo_hip_ba_handle h);
int mvo_hip_ba_partial_decide(mvo_hip_ba_handle h, int mode, int32_t *done);
int mvo_hip_ba_partial_exchange(mvo_hip_ba_handle h, int which, void *device_buffer, uint64_t n_doubles,
                                int direction);

int mvo_hip_ba_set_trace(mvo_hip_ba_handle h, int32_t capacity);
int mvo_hip_ba_get_trace(mvo_hip_ba_handle h, mvo_hip_ba_iteration *rows, int32_t capacity, int32_t *n_rows);
int mvo_hip_ba_set_profiling(mvo_hip_ba_handle h, int enabled);
int mvo_hip_ba_get_timing(mvo_hip_ba_handle h, mvo_hip_ba_timing *out);

#ifdef __cplusplus
}
#endif
#endif /* MVO_HIP_BA_H */

/* mvo_hip_epipolar.h -- C ABI of the RANSAC baseline-direction stage.
 *
 * Replaces EpipolarGeometry::estimateBaseLine and ::calculateBaseLine
 * (/root/reference/mvo/src/algorithms/EpipolarGeometry.cpp:34-134, 136-176), called once per frame
 * from BaselineEstimator::stage (mvo/src/pipeline/BaselineEstimator.cpp:46-50) two stages before
 * the refiner.  Correspondence gathering, un-rotation (BaselineEstimator.cpp:33-41), the outlier
 * bookkeeping (:51-69) and the sign vote (:72-121) stay on the caller's side.
 *
 * What runs on the GPU: every hypothesis at once -- the 3-sample least-squares direction
 * (calculateBaseLine on the drawn triple, :80) and its score over ALL correspondences (pInlier,
 * :88-101: inlier count and summed probability).  The reference's sequential bookkeeping (adaptive
 * iteration count :104-110, strict-improvement best :113-118) is then replayed over those scores
 * in draw order, so the result is the one the reference loop would reach with the same draws.
 *
 * The draws are an INPUT: the reference seeds std::mt19937 from std::random_device
 * (EpipolarGeometry.hpp:18-19) and is not reproducible; the caller passes the indices
 * randomDist(_randomGenerator) would return (three per hypothesis, with replacement, :74-78).
 * The sign of the returned direction is that of the smallest eigenvector as the library's Jacobi
 * iteration produces it; cv::SVD pins none either and the caller votes on it.
 * Rays are (x, y, 1) as the reference holds them; calculateBaseLine reads x and y only.
 */
#ifndef MVO_HIP_EPIPOLAR_H
#define MVO_HIP_EPIPOLAR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvo_hip_epipolar_context *mvo_hip_epipolar_handle;

typedef struct {
  double b[3];              /* calculateBaseLine over the best hypothesis' inliers (:122-133) */
  double best_probability;  /* pBest */
  uint32_t n_inliers;       /* size of the returned inlier index list */
  uint32_t iterations;      /* hypotheses consumed = `iteration` when the loop ends */
  uint32_t planned_iterations; /* nIterations when the loop ends */
  uint32_t best_hypothesis; /* index of the winning draw */
  int32_t exhausted;        /* 1: the adaptive loop wanted more hypotheses than draws were supplied */
  int32_t reserved;
} mvo_hip_epipolar_result; /* 56 bytes */

/* status codes are those of mvo_hip.h (0 = ok) */
int mvo_hip_epipolar_create(int device_ordinal, mvo_hip_epipolar_handle *out);
void mvo_hip_epipolar_destroy(mvo_hip_epipolar_handle h);

/* estimateBaseLine(mhi, mt, inlierIndexes, ps, threshold) with the draws supplied.
 * mhi, mt: 3 doubles per correspondence (host); samples: 3 * n_samples indices < n;
 * inlier_indexes: room for n entries, ascending on return.  n < 3 returns b = 0 (:44-48). */
int mvo_hip_epipolar_estimate(mvo_hip_epipolar_handle h, uint32_t n, const double *mhi, const double *mt,
                              uint32_t n_samples, const uint32_t *samples, double ps, double threshold,
                              uint32_t *inlier_indexes, mvo_hip_epipolar_result *result);

/* parity/debug: direction (3), inlier count and summed probability of every supplied hypothesis */
int mvo_hip_epipolar_score(mvo_hip_epipolar_handle h, uint32_t n, const double *mhi, const double *mt,
                           uint32_t n_samples, const uint32_t *samples, double threshold, double *hyp_b,
                           uint32_t *hyp_inliers, double *hyp_probability);

/* seconds the device spent scoring in the last call (HIP events) and hypotheses x correspondences */
int mvo_hip_epipolar_last_timing(mvo_hip_epipolar_handle h, double *device_seconds, uint64_t *evaluations);

#ifdef __cplusplus
}
#endif
#endif /* MVO_HIP_EPIPOLAR_H */

/* mvo_epipolar_oracle.c -- CPU restatement of EpipolarGeometry::estimateBaseLine and
 * ::calculateBaseLine (/root/reference/mvo/src/algorithms/EpipolarGeometry.cpp:11-176).
 * TEST INFRASTRUCTURE ONLY.
 *
 * Pinned BY FORMULA (the reference has no tests or fixtures for this stage and needs OpenCV, boost
 * and ROS to build): tests/golden/epipolar_*.npz hold the same quantities from an independent
 * numpy restatement that takes the null direction from numpy.linalg.svd.  Two things no
 * restatement can pin: (1) the sign of the returned baseline -- cv::SVD's sign convention is an
 * implementation detail and the reference's caller votes on it anyway (BaselineEstimator.cpp:
 * 76-121); (2) the random draws -- the reference seeds a std::mt19937 from std::random_device
 * (EpipolarGeometry.hpp:18-19), so the draws are an INPUT here (`samples`, three indices per
 * hypothesis in draw order, EpipolarGeometry.cpp:74-78).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double det3(const double *a) { /* cv::determinant(Matx33d) */
  return a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

/* unit eigenvector of the smallest eigenvalue of the symmetric matrix A (= last column of V of
 * cv::SVD for a symmetric positive semi-definite A, up to sign): two-sided Jacobi */
static void null_direction(const double *A, double *out) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(a, A, sizeof a);
  a[3] = a[1]; a[6] = a[2]; a[7] = a[5];
  static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
    if (off <= 1e-19 * (fabs(a[0]) + fabs(a[4]) + fabs(a[8])) || off < 1e-300) break;
    for (int r = 0; r < 3; ++r) {
      const int p = P[r], q = Q[r];
      const double apq = a[3 * p + q];
      if (apq == 0.0) continue;
      const double tau = (a[4 * q] - a[4 * p]) / (2.0 * apq);
      const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
      for (int k = 0; k < 3; ++k) {
        const double x = a[3 * k + p], y = a[3 * k + q];
        a[3 * k + p] = c * x - s * y;
        a[3 * k + q] = s * x + c * y;
      }
      for (int k = 0; k < 3; ++k) {
        const double x = a[3 * p + k], y = a[3 * q + k];
        a[3 * p + k] = c * x - s * y;
        a[3 * q + k] = s * x + c * y;
      }
      for (int k = 0; k < 3; ++k) {
        const double x = v[3 * k + p], y = v[3 * k + q];
        v[3 * k + p] = c * x - s * y;
        v[3 * k + q] = s * x + c * y;
      }
    }
  }
  int m = 0;
  if (a[4] < a[4 * m]) m = 1;
  if (a[8] < a[4 * m]) m = 2;
  const double n = sqrt(v[m] * v[m] + v[3 + m] * v[3 + m] + v[6 + m] * v[6 + m]);
  out[0] = v[m] / n;
  out[1] = v[3 + m] / n;
  out[2] = v[6 + m] / n;
}

/* EpipolarGeometry::calculateBaseLine(mhi, mt) over the index list idx (n entries): lines 136-176.
 * first = the array passed as the FIRST argument (m1 in the source), second = the SECOND (m2). */
void mvo_epipolar_oracle_baseline(const double *first, const double *second, const uint32_t *idx, uint32_t n,
                                  double *b) {
  double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t k = 0; k < n; ++k) {
    const double *m1 = first + 3 * (size_t)idx[k], *m2 = second + 3 * (size_t)idx[k];
    const double x1 = m1[0], y1 = m1[1], x2 = m2[0], y2 = m2[1];
    const double a1 = -(y2 - y1), a2 = (x2 - x1), a3 = ((y2 - y1) * x1 - (x2 - x1) * y2);
    A[0] += a1 * a1; A[1] += a1 * a2; A[2] += a1 * a3;
    A[3] += a2 * a1; A[4] += a2 * a2; A[5] += a2 * a3;
    A[6] += a3 * a1; A[7] += a3 * a2; A[8] += a3 * a3;
  }
  b[0] = b[1] = b[2] = 0.0; /* cv::Vec3d b; stays zero for a singular A (line 164) */
  if (det3(A) != 0) null_direction(A, b);
}

/* boost::math::binomial_coefficient<double>(n, k) for k = 3, then the source's conversion to
 * unsigned int (exact while C(n, 3) < 2^32, i.e. n <= 2953; beyond that the reference's own
 * conversion is undefined behaviour) */
static unsigned int binom3(unsigned int n) {
  if (n < 3) return 0;
  const double v = (double)n * (double)(n - 1) * (double)(n - 2) / 6.0;
  return (unsigned int)(unsigned long long)v;
}

static unsigned int iterations_for(unsigned int N, unsigned int nInlier, double ps) { /* lines 11-32 */
  const unsigned int m = binom3(nInlier), n = binom3(N);
  const double p = (double)m / (double)n;
  const double it = ceil(log(1 - ps) / log(1 - p));
  if (!(it >= 0.0)) return 0;
  if (it >= 4294967295.0) return 4294967295u;
  return (unsigned int)it;
}

double mvo_epipolar_oracle_pinlier(const double *m1, const double *m2, const double *b) { /* lines 89-94 */
  const double m1b = m1[0] * b[0] + m1[1] * b[1] + m1[2] * b[2];
  const double m2b = m2[0] * b[0] + m2[1] * b[1] + m2[2] * b[2];
  /* m2^T (I - b b^T) m1 */
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = m1[r] - b[r] * m1b;
  const double num = m2[0] * t[0] + m2[1] * t[1] + m2[2] * t[2];
  const double n1 = m1[0] * m1[0] + m1[1] * m1[1] + m1[2] * m1[2];
  const double n2 = m2[0] * m2[0] + m2[1] * m2[1] + m2[2] * m2[2];
  return num / (sqrt(n1 - pow(m1b, 2)) * sqrt(n2 - pow(m2b, 2)));
}

/* EpipolarGeometry::estimateBaseLine (lines 34-134) with the draws supplied.  Outputs: b, the best
 * hypothesis' inlier indices (ascending, as pushed), counters; per-hypothesis b / inlier count /
 * summed probability for every hypothesis actually consumed when those arrays are non-NULL.
 * Returns 0, or 1 if the adaptive loop wanted more draws than n_samples. */
int mvo_epipolar_oracle_estimate(uint32_t N, const double *mhi, const double *mt, uint32_t n_samples,
                                 const uint32_t *samples, double ps, double threshold, double *b_out,
                                 uint32_t *inliers_out, uint32_t *n_inliers, uint32_t *iterations,
                                 uint32_t *planned, uint32_t *best_hyp, double *best_prob, double *hyp_b,
                                 uint32_t *hyp_inl, double *hyp_prob) {
  const unsigned int s = 3;
  double inlierProbability = 0.3;
  b_out[0] = b_out[1] = b_out[2] = 0.0;
  *n_inliers = *iterations = *planned = *best_hyp = 0;
  *best_prob = -1;
  if (N < 3) return 0; /* lines 44-48 */
  if (inlierProbability * N < s) inlierProbability = 1;
  unsigned int nIterations = iterations_for(N, (unsigned int)(int)(N * inlierProbability), ps);
  unsigned int iteration = 0;
  double pBest = -1;
  uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * N), *best = (uint32_t *)malloc(sizeof(uint32_t) * N);
  uint32_t nBestIdx = 0;
  int exhausted = 0;
  while (iteration < nIterations) {
    if (iteration >= n_samples) {
      exhausted = 1;
      break;
    }
    double b[3];
    /* calculateBaseLine(mtSet, mhiSet): first argument = mt, second = mhi (line 80) */
    mvo_epipolar_oracle_baseline(mt, mhi, samples + 3 * (size_t)iteration, 3, b);
    unsigned int nInlier = 0;
    double nProbability = 0;
    for (uint32_t i = 0; i < N; ++i) {
      const double pInlier = mvo_epipolar_oracle_pinlier(mhi + 3 * (size_t)i, mt + 3 * (size_t)i, b);
      if (pInlier > threshold) {
        cur[nInlier++] = i;
        nProbability += pInlier;
      }
    }
    if (hyp_b) memcpy(hyp_b + 3 * (size_t)iteration, b, sizeof b);
    if (hyp_inl) hyp_inl[iteration] = nInlier;
    if (hyp_prob) hyp_prob[iteration] = nProbability;
    iteration++;
    if ((double)nInlier / (double)N > inlierProbability) {
      inlierProbability = (double)nInlier / (double)N;
      nIterations = iterations_for(N, nInlier, ps);
    }
    if (nProbability > pBest) {
      pBest = nProbability;
      nBestIdx = nInlier;
      memcpy(best, cur, sizeof(uint32_t) * nInlier);
      *best_hyp = iteration - 1;
    }
  }
  for (uint32_t k = 0; k < nBestIdx; ++k) inliers_out[k] = best[k];
  *n_inliers = nBestIdx;
  *iterations = iteration;
  *planned = nIterations;
  *best_prob = pBest;
  mvo_epipolar_oracle_baseline(mt, mhi, best, nBestIdx, b_out); /* lines 122-133 */
  free(cur);
  free(best);
  return exhausted;
}

/* every supplied hypothesis scored against every correspondence (lines 80-101 without the adaptive
 * bookkeeping): the CPU counterpart of mvo_hip_epipolar_score */
void mvo_epipolar_oracle_score(uint32_t N, const double *mhi, const double *mt, uint32_t K, const uint32_t *samples,
                               double threshold, double *hyp_b, uint32_t *hyp_inl, double *hyp_prob) {
  for (uint32_t k = 0; k < K; ++k) {
    double b[3];
    mvo_epipolar_oracle_baseline(mt, mhi, samples + 3 * (size_t)k, 3, b);
    unsigned int nInlier = 0;
    double nProbability = 0;
    for (uint32_t i = 0; i < N; ++i) {
      const double pInlier = mvo_epipolar_oracle_pinlier(mhi + 3 * (size_t)i, mt + 3 * (size_t)i, b);
      if (pInlier > threshold) {
        nInlier++;
        nProbability += pInlier;
      }
    }
    if (hyp_b) memcpy(hyp_b + 3 * (size_t)k, b, sizeof b);
    hyp_inl[k] = nInlier;
    hyp_prob[k] = nProbability;
  }
}

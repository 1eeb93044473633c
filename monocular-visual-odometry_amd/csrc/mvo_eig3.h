// mvo_eig3.h -- smallest-eigenvalue eigenvector of a symmetric 3x3 matrix by cyclic Jacobi
// rotations (host and device).  For the symmetric positive semi-definite A = sum a a^T of
// EpipolarGeometry::calculateBaseLine (/root/reference/mvo/src/algorithms/EpipolarGeometry.cpp:
// 142-172) this is the last right singular vector cv::SVD returns -- up to its sign, which no SVD
// pins and which the reference's caller settles by a vote (BaselineEstimator.cpp:76-121).
#pragma once

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define MVO_HD __host__ __device__
#else
#define MVO_HD
#endif
#include <math.h>

// A: row-major 3x3 (only the symmetric part is used); out: unit eigenvector of the smallest eigenvalue
MVO_HD static inline void mvo_eig3_smallest(const double* A, double out[3]) {
  double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[4], A[5]}, {A[2], A[5], A[8]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-18 * dia) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A J
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- J^T A
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int m = 0;
  if (a[1][1] < a[m][m]) m = 1;
  if (a[2][2] < a[m][m]) m = 2;
  const double nrm = sqrt(v[0][m] * v[0][m] + v[1][m] * v[1][m] + v[2][m] * v[2][m]);
  out[0] = v[0][m] / nrm;
  out[1] = v[1][m] / nrm;
  out[2] = v[2][m] / nrm;
}

// cv::determinant of a Matx33d (the explicit 3x3 cofactor expansion OpenCV uses for small matrices)
MVO_HD static inline double mvo_det3(const double* a) {
  return a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

// one row of EpipolarGeometry::calculateBaseLine (EpipolarGeometry.cpp:146-152): m1 = first
// argument's ray, m2 = second argument's ray; only x and y are used (rays carry z = 1)
MVO_HD static inline void mvo_epi_row(const double* m1, const double* m2, double a[3]) {
  const double x1 = m1[0], y1 = m1[1], x2 = m2[0], y2 = m2[1];
  a[0] = -(y2 - y1);
  a[1] = (x2 - x1);
  a[2] = ((y2 - y1) * x1 - (x2 - x1) * y2);
}

// pInlier of EpipolarGeometry.cpp:90-94
MVO_HD static inline double mvo_epi_pinlier(const double* m1, const double* m2, const double* b) {
  // every operation rounded by itself, as the reference's host build does (x86-64 without FMA): a
  // contracted multiply-add would move p by an ulp, and `p > threshold` feeds an integer count
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double m1b = m1[0] * b[0] + m1[1] * b[1] + m1[2] * b[2];
  const double m2b = m2[0] * b[0] + m2[1] * b[1] + m2[2] * b[2];
  const double m12 = m1[0] * m2[0] + m1[1] * m2[1] + m1[2] * m2[2];
  const double n1 = m1[0] * m1[0] + m1[1] * m1[1] + m1[2] * m1[2];
  const double n2 = m2[0] * m2[0] + m2[1] * m2[1] + m2[2] * m2[2];
  return (m12 - m2b * m1b) / (sqrt(n1 - m1b * m1b) * sqrt(n2 - m2b * m2b));
}

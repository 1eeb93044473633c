/* mvo_scaler_oracle.c -- CPU restatement of Scaler::stage's arithmetic
 * (/root/reference/mvo/src/pipeline/Scaler.cpp:34-67).  TEST INFRASTRUCTURE ONLY.
 * The reference has no tests or fixtures for this stage and cannot be built here (OpenCV / ROS
 * headers are absent), so parity is pinned BY FORMULA: this file follows the source expression by
 * expression, and tests/golden/scaler_*.npz hold the same quantities from an independent numpy
 * restatement (tests/golden/make_scaler_golden.py). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void matvec(const double *R, const double *v, double s, double *o) {
  for (int r = 0; r < 3; ++r) o[r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2]) / s;
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double norm3(const double *a) { return sqrt(dot3(a, a)); }

/* Scaler.cpp:35-43 */
double mvo_scaler_oracle_track(const double *R0, const double *R1, const double *R2, const double *u1, double n1,
                               const double *v0, const double *v1, const double *v2) {
  double m0[3], m1[3], m2[3];
  matvec(R0, v0, norm3(v0), m0);
  matvec(R1, v1, norm3(v1), m1);
  matvec(R2, v2, norm3(v2), m2);
  const double counting = n1 * sqrt(1.0 - pow(dot3(m2, u1), 2)) * sqrt(1.0 - pow(dot3(m1, m0), 2));
  const double naming = sqrt(1.0 - pow(dot3(m2, m1), 2)) * sqrt(1.0 - pow(dot3(u1, m0), 2));
  return counting / naming;
}

static int cmp_nan_last(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  if (isnan(x)) return isnan(y) ? 0 : 1;
  if (isnan(y)) return -1;
  return (x > y) - (x < y);
}

/* Scaler.cpp:48-67; returns the number of tracks; scales (n doubles) receives the per-track values
 * in input order when non-NULL */
uint64_t mvo_scaler_oracle_run(const double *R0, const double *R1, const double *R2, const double *u1, double n1,
                               double lowest, double highest, uint64_t n, const double *v0, const double *v1,
                               const double *v2, double *scales, double *median, double *scale, int32_t *clamped) {
  if (n == 0) return 0;
  double *s = (double *)malloc(sizeof(double) * n);
  for (uint64_t i = 0; i < n; ++i) {
    s[i] = mvo_scaler_oracle_track(R0, R1, R2, u1, n1, v0 + 3 * i, v1 + 3 * i, v2 + 3 * i);
    if (scales) scales[i] = s[i];
  }
  qsort(s, n, sizeof(double), cmp_nan_last);
  double n0;
  if (n % 2 == 0) n0 = (s[(n - 1) / 2] + s[(n - 1) / 2 + 1]) / 2;
  else n0 = s[(n - 1) / 2];
  free(s);
  *median = n0;
  *clamped = 0;
  if (n0 > highest) {
    *scale = highest;
    *clamped = 1;
  } else if (n0 < lowest) {
    *scale = lowest;
    *clamped = -1;
  } else {
    *scale = n0;
  }
  return n;
}

/*
 * mvo_oracle.c -- CPU restatement of the reference's sliding-window refinement.
 * TEST INFRASTRUCTURE ONLY (see mvo_oracle.h for scope, citations and parity status:
 * residual/Jacobian pinned by formula; LM trajectory "parity unpinned" -- Ceres absent).
 *
 * Reference = /root/reference/mvo/src/algorithms/IterativeRefinement.cpp (cited as IR.cpp).
 */
#include "mvo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXM 5            /* IR.cpp:384-451: switch over 1..5 free frames, default: assert(false) */
#define MAXC (3 * MAXM)   /* local columns: per refined frame (da, db, dt) */
#define MAXG (4 * MAXM)   /* global parameters: per refined frame vec(3) + t(1) */

/* Threads for the timing legs of bench.py's cpu_baseline (1 thread, the reference's
 * default of 4 = maxNumThreads, mvo/cfg/mvo.cfg:47, all cores).  Ceres 1.14 evaluates residual blocks
 * on num_threads threads; the dense QR is parallelised here as well (row ranges reduced separately,
 * then combined), which is more than Ceres does -- a generous baseline.  Default 1: every parity test
 * runs the serial path, whose summation order never changes. */
static int g_threads = 1;
void mvo_oracle_set_threads(int32_t n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
int32_t mvo_oracle_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------------------ */
/* scalar helpers                                                                        */
/* ------------------------------------------------------------------------------------ */

/* IR.cpp:298-309 scaleTemplated<double>. */
double mvo_oracle_scale(double t, double max_len, double min_len) {
  double e = exp(-1.0 * t);
  if (isinf(e)) return min_len;
  return min_len + ((max_len - min_len) / (1.0 + e));
}

/* d/dt of the above as Ceres' Jet arithmetic produces it: exp(-t) -> (e, -e);
 * c/(1+e) -> derivative  (-c/((1+e)(1+e))) * (-e).  The Inf branch returns the constant
 * T(MIN_LEN), i.e. derivative 0; for huge finite e the quotient underflows to 0 as well. */
double mvo_oracle_scale_derivative(double t, double max_len, double min_len) {
  double e = exp(-1.0 * t);
  if (isinf(e)) return 0.0;
  double g = 1.0 + e;
  return (-(max_len - min_len) / (g * g)) * (-e);
}

/* IR.cpp:348-368 reverseScale (the ROS_WARN logging is not part of the arithmetic). */
double mvo_oracle_reverse_scale(double length, double max_len, double min_len) {
  if (length <= min_len) {
    return mvo_oracle_reverse_scale(min_len + (length * DBL_EPSILON) + DBL_EPSILON, max_len,
                                    min_len);
  } else if (length >= max_len) {
    return mvo_oracle_reverse_scale(max_len - (length * DBL_EPSILON + DBL_EPSILON), max_len,
                                    min_len);
  }
  return -1.0 * log((max_len - length) / (length - min_len));
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

static void mat3_vec(const double A[9], const double x[3], double y[3]) {
  for (int i = 0; i < 3; i++) y[i] = A[3 * i + 0] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}

static void mat3t_vec(const double A[9], const double x[3], double y[3]) {
  for (int j = 0; j < 3; j++) y[j] = A[0 + j] * x[0] + A[3 + j] * x[1] + A[6 + j] * x[2];
}

static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

static double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* IR.cpp:282-296 baseLineTemplated + :334-346 ParametrizedBaseLine: x (+) (a,b) = (B*A)*x. */
void mvo_oracle_plus(const double x[3], const double delta[2], double out[3]) {
  const double a = delta[0], b = delta[1];
  const double A[9] = {(1.0 - a * a) / (1.0 + a * a), (-2.0 * a) / (1.0 + a * a), 0.0,
                       (2.0 * a) / (1.0 + a * a), (1.0 - a * a) / (1.0 + a * a), 0.0,
                       0.0, 0.0, 1.0};
  const double B[9] = {(1.0 - b * b) / (1.0 + b * b), 0.0, (2.0 * b) / (1.0 + b * b),
                       0.0, 1.0, 0.0,
                       (-2.0 * b) / (1.0 + b * b), 0.0, (1.0 - b * b) / (1.0 + b * b)};
  double BA[9];
  mat3_mul(B, A, BA);
  mat3_vec(BA, x, out);
}

/* operations/FeatureOperations.cpp:117-122 normFeatures. */
void mvo_oracle_norm_features(double *rays, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    double *f = rays + 3 * i;
    double scale = 1.0 / sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    f[0] *= scale;
    f[1] *= scale;
    f[2] *= scale;
  }
}

/* ------------------------------------------------------------------------------------ */
/* pair enumeration: IR.cpp:116 (frame0 < numberToRefine), :134 (frame0 <= frame1 < N-1)  */
/* ------------------------------------------------------------------------------------ */
int32_t mvo_oracle_num_pairs(uint32_t M, uint32_t N) {
  if (M == 0 || N <= M) return -1;
  return (int32_t)(M * (N - 1) - (M * (M - 1)) / 2);
}

void mvo_oracle_pair(uint32_t M, uint32_t N, uint32_t idx, uint32_t *a, uint32_t *b) {
  uint32_t k = 0;
  for (uint32_t f0 = 0; f0 < M; f0++)
    for (uint32_t f1 = f0; f1 < N - 1; f1++, k++)
      if (k == idx) {
        *a = f0;
        *b = f1;
        return;
      }
  *a = *b = 0xffffffffu;
}

/* ------------------------------------------------------------------------------------ */
/* problem context                                                                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  const mvo_oracle_frame *frames;
  uint32_t M, N;
  const mvo_oracle_options *opt;
  const uint64_t *off;
  const double *m0, *m1;
  uint32_t n_pairs;
  uint64_t n_obs;
  int ncol;          /* active local columns: 3M, or 2M when fix_length */
  int colmap[MAXC];  /* full column (3k+j) -> active column or -1 */
} ctx_t;

static int ctx_init(ctx_t *c, const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                    const mvo_oracle_options *opt, const uint64_t *off, const double *m0,
                    const double *m1) {
  if (M == 0 || M > MAXM || N <= M) return -1; /* IR.cpp:28-30 asserts; :450 assert(false) */
  c->frames = frames;
  c->M = M;
  c->N = N;
  c->opt = opt;
  c->off = off;
  c->m0 = m0;
  c->m1 = m1;
  c->n_pairs = (uint32_t)mvo_oracle_num_pairs(M, N);
  c->n_obs = off[c->n_pairs];
  int n = 0;
  for (uint32_t k = 0; k < M; k++) {
    c->colmap[3 * k + 0] = n++;
    c->colmap[3 * k + 1] = n++;
    /* IR.cpp:105-107: fixLength => SetParameterBlockConstant(scale) => column removed */
    c->colmap[3 * k + 2] = (opt && opt->fix_length) ? -1 : n++;
  }
  c->ncol = n;
  return 0;
}

/* CauchyLoss(a) as published by Ceres: b = a^2, c = 1/b;
 * rho0 = b log(1+s c), rho1 = max(DBL_MIN, 1/(1+s c)), rho2 = -c/(1+s c)^2 (always < 0). */
static void cauchy(double s, double rho[3]) {
  const double b = 0.5 * 0.5, c = 1.0 / b;
  const double sum = 1.0 + s * c;
  const double inv = 1.0 / sum;
  rho[0] = b * log(sum);
  rho[1] = inv > DBL_MIN ? inv : DBL_MIN;
  rho[2] = -c * (inv * inv);
}

/*
 * Evaluate cost (and optionally residuals / Jacobian / gradient) at the point
 * (vecs[k][3], t[k]), k < M.
 *   raw != 0  : no loss correction on residuals / Jacobian (debug outputs), all 3M columns
 *   raw == 0  : Ceres' ResidualBlock::Evaluate semantics -- cost = 1/2 sum rho(r^2); since
 *               rho'' < 0 for Cauchy the Corrector scales residual and Jacobian row by
 *               sqrt(rho'); only active columns are emitted (jac is n_obs x ncol).
 */
/* one frame pair (a, b) = pair index p: adds to *cost_acc and grad_acc (may be NULL) */
static void evaluate_pair(const ctx_t *c, const double vecs[][3], const double *s, const double *sp, int raw,
                          int use_loss, int ncol, uint32_t p, uint32_t a, uint32_t b, double *res, double *jac,
                          double *cost_acc, double *grad_acc) {
  const uint32_t M = c->M;
  double cost = *cost_acc;
  const uint32_t kmax = b < M - 1 ? b : M - 1;
  /* IR.cpp:137-144: fresh zero offset per pair; only frame b itself is added if b >= M */
  double offset[3] = {0.0, 0.0, 0.0};
  if (b >= M)
    for (int j = 0; j < 3; j++) offset[j] = c->frames[b].scale * c->frames[b].vec[j];
  /* IR.cpp:466-577: s0*v0 + s1*v1 + ... summed left to right */
  double u[3] = {0.0, 0.0, 0.0};
  for (uint32_t k = a; k <= kmax; k++)
    for (int j = 0; j < 3; j++) {
      double term = s[k] * vecs[k][j];
      u[j] = (k == a) ? term : u[j] + term;
    }
  /* IR.cpp:254-255 */
  for (int j = 0; j < 3; j++) u[j] = u[j] + offset[j];
  const double nb = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  for (int j = 0; j < 3; j++) u[j] /= nb;

  const double *R0 = c->frames[a].R;      /* IR.cpp:172 */
  const double *R1 = c->frames[b + 1].R;  /* IR.cpp:171 */

  for (uint64_t i = c->off[p]; i < c->off[p + 1]; i++) {
    const double *m0 = c->m0 + 3 * i, *m1 = c->m1 + 3 * i;
    double p0[3], x[3], y[3];
    mat3_vec(R0, m0, p0);
    cross3(u, p0, x);
    mat3t_vec(R1, x, y);
    double r = dot3(m1, y); /* IR.cpp:257-258 */
    double sr = 1.0;
    if (use_loss) {
      double rho[3];
      cauchy(r * r, rho);
      cost += 0.5 * rho[0];
      sr = sqrt(rho[1]);
    } else {
      cost += 0.5 * (r * r);
    }
    if (jac || grad_acc) {
      double row[MAXC];
      for (int j = 0; j < ncol; j++) row[j] = 0.0;
      double p1[3], cc[3], g[3];
      mat3_vec(R1, m1, p1);
      cross3(p0, p1, cc);
      const double uc = dot3(u, cc);
      for (int j = 0; j < 3; j++) g[j] = (cc[j] - u[j] * uc) / nb;
      for (uint32_t k = a; k <= kmax; k++) {
        const double *v = vecs[k];
        const double jl0[3] = {-2.0 * v[1], 2.0 * v[0], 0.0};
        const double jl1[3] = {2.0 * v[2], 0.0, -2.0 * v[0]};
        const double d0 = s[k] * dot3(g, jl0);
        const double d1 = s[k] * dot3(g, jl1);
        const double dt = sp[k] * dot3(v, g);
        if (raw) {
          row[3 * k + 0] = d0;
          row[3 * k + 1] = d1;
          row[3 * k + 2] = dt;
        } else {
          row[c->colmap[3 * k + 0]] = d0 * sr;
          row[c->colmap[3 * k + 1]] = d1 * sr;
          if (c->colmap[3 * k + 2] >= 0) row[c->colmap[3 * k + 2]] = dt * sr;
        }
      }
      if (jac) memcpy(jac + (size_t)i * ncol, row, sizeof(double) * ncol);
      if (grad_acc)
        for (int j = 0; j < ncol; j++) grad_acc[j] += row[j] * (r * sr);
    }
    if (res) res[i] = r * sr;
  }
  *cost_acc = cost;
}

static double evaluate(const ctx_t *c, const double vecs[][3], const double *t, int raw,
                       double *res, double *jac, double *grad) {
  const uint32_t M = c->M, N = c->N;
  const double MAX = c->opt->highest_length, MIN = c->opt->lowest_length;
  const int use_loss = c->opt->use_loss_function && !raw;
  const int ncol = raw ? (int)(3 * M) : c->ncol;
  double cost = 0.0;
  if (grad)
    for (int j = 0; j < ncol; j++) grad[j] = 0.0;

  double s[MAXM], sp[MAXM];
  for (uint32_t k = 0; k < M; k++) {
    s[k] = mvo_oracle_scale(t[k], MAX, MIN);
    sp[k] = mvo_oracle_scale_derivative(t[k], MAX, MIN);
  }

  if (g_threads <= 1) { /* the parity path: one running sum over all residual blocks, pair-major */
    uint32_t p = 0;
    for (uint32_t a = 0; a < M; a++)
      for (uint32_t b = a; b < N - 1; b++, p++)
        evaluate_pair(c, vecs, s, sp, raw, use_loss, ncol, p, a, b, res, jac, &cost, grad);
    return cost;
  }

  /* timing legs: pairs in parallel, per-pair sums added up in pair order afterwards */
  const int P = (int)c->n_pairs;
  double *pc = (double *)calloc((size_t)P * (MAXC + 1), sizeof(double));
  if (!pc) return NAN;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
  for (int p = 0; p < P; p++) {
    uint32_t a, b;
    mvo_oracle_pair(M, N, (uint32_t)p, &a, &b);
    evaluate_pair(c, vecs, s, sp, raw, use_loss, ncol, (uint32_t)p, a, b, res, jac, pc + (size_t)p * (MAXC + 1),
                  grad ? pc + (size_t)p * (MAXC + 1) + 1 : NULL);
  }
  for (int p = 0; p < P; p++) {
    cost += pc[(size_t)p * (MAXC + 1)];
    if (grad)
      for (int j = 0; j < ncol; j++) grad[j] += pc[(size_t)p * (MAXC + 1) + 1 + j];
  }
  free(pc);
  return cost;
}

static void initial_point(const ctx_t *c, double vecs[][3], double *t) {
  /* IR.cpp:95-109 */
  for (uint32_t k = 0; k < c->M; k++) {
    for (int j = 0; j < 3; j++) vecs[k][j] = c->frames[k].vec[j];
    t[k] = mvo_oracle_reverse_scale(c->frames[k].scale, c->opt->highest_length,
                                    c->opt->lowest_length);
  }
}

int32_t mvo_oracle_eval(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                        const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                        const double *m0, const double *m1, double *residuals,
                        double *jacobian, double *cost) {
  ctx_t c;
  if (ctx_init(&c, frames, M, N, opt, pair_offsets, m0, m1)) return -1;
  double vecs[MAXM][3], t[MAXM];
  initial_point(&c, vecs, t);
  double v = evaluate(&c, (const double(*)[3])vecs, t, 1, residuals, jacobian, NULL);
  if (cost) *cost = v;
  return 0;
}

int32_t mvo_oracle_eval_at(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                           const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                           const double *m0, const double *m1, const double *point, double *residuals,
                           double *jacobian, double *cost) {
  ctx_t c;
  if (ctx_init(&c, frames, M, N, opt, pair_offsets, m0, m1)) return -1;
  double vecs[MAXM][3], t[MAXM];
  initial_point(&c, vecs, t);
  if (point)
    for (uint32_t k = 0; k < M; k++) {
      for (int j = 0; j < 3; j++) vecs[k][j] = point[4 * k + j];
      t[k] = point[4 * k + 3];
    }
  double v = evaluate(&c, (const double(*)[3])vecs, t, 1, residuals, jacobian, NULL);
  if (cost) *cost = v;
  return 0;
}

int32_t mvo_oracle_pair_moments(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                                const uint64_t *off, const double *m0, const double *m1,
                                double *moments) {
  if (M == 0 || M > MAXM || N <= M) return -1;
  uint32_t p = 0;
  for (uint32_t a = 0; a < M; a++)
    for (uint32_t b = a; b < N - 1; b++, p++) {
      double S[6] = {0, 0, 0, 0, 0, 0};
      for (uint64_t i = off[p]; i < off[p + 1]; i++) {
        double p0[3], p1[3], c[3];
        mat3_vec(frames[a].R, m0 + 3 * i, p0);
        mat3_vec(frames[b + 1].R, m1 + 3 * i, p1);
        cross3(p0, p1, c);
        S[0] += c[0] * c[0];
        S[1] += c[0] * c[1];
        S[2] += c[0] * c[2];
        S[3] += c[1] * c[1];
        S[4] += c[1] * c[2];
        S[5] += c[2] * c[2];
      }
      memcpy(moments + 6 * (size_t)p, S, sizeof S);
    }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* forward-mode dual numbers through the reference's own expression tree                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  double a;
  double v[MAXG];
} jet;

static jet jconst(double a) {
  jet r;
  r.a = a;
  memset(r.v, 0, sizeof r.v);
  return r;
}
static jet jvar(double a, int slot) {
  jet r = jconst(a);
  r.v[slot] = 1.0;
  return r;
}
static jet jadd(jet x, jet y) {
  jet r;
  r.a = x.a + y.a;
  for (int i = 0; i < MAXG; i++) r.v[i] = x.v[i] + y.v[i];
  return r;
}
static jet jsub(jet x, jet y) {
  jet r;
  r.a = x.a - y.a;
  for (int i = 0; i < MAXG; i++) r.v[i] = x.v[i] - y.v[i];
  return r;
}
static jet jmul(jet x, jet y) {
  jet r;
  r.a = x.a * y.a;
  for (int i = 0; i < MAXG; i++) r.v[i] = x.a * y.v[i] + x.v[i] * y.a;
  return r;
}
static jet jdiv(jet x, jet y) {
  jet r;
  const double inv = 1.0 / y.a;
  r.a = x.a * inv;
  for (int i = 0; i < MAXG; i++) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv;
  return r;
}
static jet jscale(double s, jet x) {
  jet r;
  r.a = s * x.a;
  for (int i = 0; i < MAXG; i++) r.v[i] = s * x.v[i];
  return r;
}
static jet jsqrt(jet x) {
  jet r;
  r.a = sqrt(x.a);
  const double d = 1.0 / (2.0 * r.a);
  for (int i = 0; i < MAXG; i++) r.v[i] = x.v[i] * d;
  return r;
}
static jet jexp(jet x) {
  jet r;
  r.a = exp(x.a);
  for (int i = 0; i < MAXG; i++) r.v[i] = r.a * x.v[i];
  return r;
}
static int jisinf(jet x) {
  if (isinf(x.a)) return 1;
  for (int i = 0; i < MAXG; i++)
    if (isinf(x.v[i])) return 1;
  return 0;
}

/* IR.cpp:298-309 in jets */
static jet jet_scale(jet t, double MAX, double MIN) {
  jet e = jexp(jscale(-1.0, t));
  if (jisinf(e)) return jconst(MIN);
  jet one_plus = jadd(jconst(1.0), e);
  return jadd(jconst(MIN), jdiv(jconst(MAX - MIN), one_plus));
}

/* IR.cpp:282-296 in jets; delta slots 0 and 1 */
static void jet_plus(const double x[3], jet a, jet b, jet out[3]) {
  jet one = jconst(1.0), two = jconst(2.0), zero = jconst(0.0), mtwo = jconst(-2.0);
  jet aa = jmul(a, a), bb = jmul(b, b);
  jet ca = jdiv(jsub(one, aa), jadd(one, aa));
  jet sa = jdiv(jmul(two, a), jadd(one, aa));
  jet msa = jdiv(jmul(mtwo, a), jadd(one, aa));
  jet cb = jdiv(jsub(one, bb), jadd(one, bb));
  jet sb = jdiv(jmul(two, b), jadd(one, bb));
  jet msb = jdiv(jmul(mtwo, b), jadd(one, bb));
  jet A[9] = {ca, msa, zero, sa, ca, zero, zero, zero, one};
  jet B[9] = {cb, zero, sb, zero, one, zero, msb, zero, cb};
  jet BA[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      BA[3 * i + j] = jadd(jadd(jmul(B[3 * i], A[j]), jmul(B[3 * i + 1], A[3 + j])),
                           jmul(B[3 * i + 2], A[6 + j]));
  for (int i = 0; i < 3; i++)
    out[i] = jadd(jadd(jscale(x[0], BA[3 * i]), jscale(x[1], BA[3 * i + 1])),
                  jscale(x[2], BA[3 * i + 2]));
}

int32_t mvo_oracle_eval_autodiff(const mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                                 const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                                 const double *m0a, const double *m1a, double *residuals,
                                 double *jacobian, double *cost_out) {
  ctx_t c;
  if (ctx_init(&c, frames, M, N, opt, pair_offsets, m0a, m1a)) return -1;
  double vecs[MAXM][3], t[MAXM];
  initial_point(&c, vecs, t);
  const double MAX = opt->highest_length, MIN = opt->lowest_length;

  /* AutoDiffLocalParameterization<ParametrizedBaseLine,3,2>: d Plus(x,delta)/d delta at 0 */
  double jloc[MAXM][3][2];
  for (uint32_t k = 0; k < M; k++) {
    jet out[3];
    jet_plus(vecs[k], jvar(0.0, 0), jvar(0.0, 1), out);
    for (int i = 0; i < 3; i++) {
      jloc[k][i][0] = out[i].v[0];
      jloc[k][i][1] = out[i].v[1];
    }
  }

  double cost = 0.0;
  uint32_t p = 0;
  for (uint32_t a = 0; a < M; a++)
    for (uint32_t b = a; b < N - 1; b++, p++) {
      const uint32_t kmax = b < M - 1 ? b : M - 1;
      double offset[3] = {0, 0, 0};
      if (b >= M)
        for (int j = 0; j < 3; j++) offset[j] = frames[b].scale * frames[b].vec[j];
      /* parameter jets: slot 4k = t_k, 4k+1..3 = vec_k */
      jet base[3];
      for (uint32_t k = a; k <= kmax; k++) {
        jet sk = jet_scale(jvar(t[k], 4 * (int)k), MAX, MIN);
        for (int j = 0; j < 3; j++) {
          jet term = jmul(sk, jvar(vecs[k][j], 4 * (int)k + 1 + j));
          base[j] = (k == a) ? term : jadd(base[j], term);
        }
      }
      jet u[3];
      for (int j = 0; j < 3; j++) u[j] = jadd(base[j], jconst(offset[j]));
      jet nrm = jsqrt(jadd(jadd(jmul(u[0], u[0]), jmul(u[1], u[1])), jmul(u[2], u[2])));
      for (int j = 0; j < 3; j++) u[j] = jdiv(u[j], nrm);
      const double *R0 = frames[a].R, *R1 = frames[b + 1].R;
      for (uint64_t i = pair_offsets[p]; i < pair_offsets[p + 1]; i++) {
        double p0[3];
        mat3_vec(R0, m0a + 3 * i, p0);
        jet x[3];
        x[0] = jsub(jscale(p0[2], u[1]), jscale(p0[1], u[2]));
        x[1] = jsub(jscale(p0[0], u[2]), jscale(p0[2], u[0]));
        x[2] = jsub(jscale(p0[1], u[0]), jscale(p0[0], u[1]));
        jet r = jconst(0.0);
        for (int j = 0; j < 3; j++) {
          jet yj = jadd(jadd(jscale(R1[0 + j], x[0]), jscale(R1[3 + j], x[1])),
                        jscale(R1[6 + j], x[2]));
          r = jadd(r, jscale(m1a[3 * i + j], yj));
        }
        cost += 0.5 * r.a * r.a;
        if (residuals) residuals[i] = r.a;
        if (jacobian) {
          double *row = jacobian + (size_t)i * 3 * M;
          for (uint32_t k = 0; k < 3 * M; k++) row[k] = 0.0;
          for (uint32_t k = a; k <= kmax; k++) {
            const double *gv = &r.v[4 * k + 1];
            row[3 * k + 0] = gv[0] * jloc[k][0][0] + gv[1] * jloc[k][1][0] + gv[2] * jloc[k][2][0];
            row[3 * k + 1] = gv[0] * jloc[k][0][1] + gv[1] * jloc[k][1][1] + gv[2] * jloc[k][2][1];
            row[3 * k + 2] = r.v[4 * k];
          }
        }
      }
    }
  if (cost_out) *cost_out = cost;
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* dense QR least squares (Ceres DenseQRSolver): min || [J; D] y - [r; 0] ||              */
/* Sequential row-block Householder (TSQR): mathematically a Householder QR of the stacked */
/* matrix; only the n x n triangle and the transformed right-hand side are kept.           */
/* ------------------------------------------------------------------------------------ */
#define QR_BLOCK 2048

static void householder_rows(double *W, int m, int n) {
  /* W is m x (n+1) row-major: columns 0..n-1 matrix, column n rhs; reduce to upper form */
  const int ld = n + 1;
  for (int j = 0; j < n && j < m; j++) {
    double nrm2 = 0.0;
    for (int i = j; i < m; i++) nrm2 += W[i * ld + j] * W[i * ld + j];
    if (nrm2 == 0.0) continue;
    double nrm = sqrt(nrm2);
    double ajj = W[j * ld + j];
    double alpha = ajj > 0 ? -nrm : nrm;
    double vj = ajj - alpha;
    double vtv = nrm2 - ajj * ajj + vj * vj;
    if (vtv == 0.0) continue;
    for (int col = j + 1; col <= n; col++) {
      double dotv = vj * W[j * ld + col];
      for (int i = j + 1; i < m; i++) dotv += W[i * ld + j] * W[i * ld + col];
      double f = 2.0 * dotv / vtv;
      W[j * ld + col] -= f * vj;
      for (int i = j + 1; i < m; i++) W[i * ld + col] -= f * W[i * ld + j];
    }
    W[j * ld + j] = alpha;
    for (int i = j + 1; i < m; i++) W[i * ld + j] = 0.0;
  }
}

/* folds rows [lo, hi) of [J | r] into the running triangle held in the top n rows of W */
static void qr_fold_rows(double *W, const double *J, const double *r, uint64_t lo, uint64_t hi, int n) {
  const int ld = n + 1;
  uint64_t done = lo;
  while (done < hi) {
    uint64_t nb = hi - done < QR_BLOCK ? hi - done : QR_BLOCK;
    for (uint64_t i = 0; i < nb; i++) {
      memcpy(W + (size_t)(n + i) * ld, J + (size_t)(done + i) * n, sizeof(double) * n);
      W[(size_t)(n + i) * ld + n] = r[done + i];
    }
    householder_rows(W, (int)(n + nb), n);
    done += nb;
  }
}

/* returns 0 ok, 1 = non-finite / singular (=> LINEAR_SOLVER_FAILURE) */
static int lsq_qr(const double *J, const double *r, uint64_t m, int n, const double *D, double *y) {
  const int ld = n + 1;
  int T = g_threads;
  if ((uint64_t)T * QR_BLOCK > m) T = 1;
  double *W = (double *)calloc((size_t)(n + QR_BLOCK) * ld * (size_t)T, sizeof(double));
  if (!W) return 1;
  /* top n rows of W hold the running [R | Q^T r]; start with the LM rows [D | 0] */
  for (int i = 0; i < n; i++) W[i * ld + i] = D[i];
  if (T == 1) {
    qr_fold_rows(W, J, r, 0, m, n);
  } else {
    /* timing legs: every thread reduces its own row range to a triangle; the T triangles are
     * stacked under the first and reduced once more */
    const size_t stride = (size_t)(n + QR_BLOCK) * ld;
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int k = 0; k < T; k++) qr_fold_rows(W + stride * k, J, r, m * k / T, m * (k + 1) / T, n);
    for (int k = 1; k < T; k++) {
      memcpy(W + (size_t)n * ld, W + stride * k, sizeof(double) * (size_t)n * ld);
      householder_rows(W, 2 * n, n);
    }
  }
  int bad = 0;
  for (int i = n - 1; i >= 0; i--) {
    double acc = W[i * ld + n];
    for (int j = i + 1; j < n; j++) acc -= W[i * ld + j] * y[j];
    y[i] = acc / W[i * ld + i];
    if (!isfinite(y[i])) bad = 1;
  }
  free(W);
  return bad;
}

/* ------------------------------------------------------------------------------------ */
/* Ceres 1.14 TrustRegionMinimizer + LevenbergMarquardtStrategy + TrustRegionStepEvaluator */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  double vecs[MAXM][3];
  double t[MAXM];
} point_t;

static void point_plus(const ctx_t *c, const point_t *x, const double *delta, point_t *out) {
  for (uint32_t k = 0; k < c->M; k++) {
    double d2[2] = {delta[c->colmap[3 * k]], delta[c->colmap[3 * k + 1]]};
    mvo_oracle_plus(x->vecs[k], d2, out->vecs[k]);
    out->t[k] = x->t[k];
    if (c->colmap[3 * k + 2] >= 0) out->t[k] = x->t[k] + delta[c->colmap[3 * k + 2]];
  }
}

/* norms over the program's (non-constant) global parameters */
static double point_diff_norm(const ctx_t *c, const point_t *a, const point_t *b, int inf_norm) {
  double acc = 0.0;
  for (uint32_t k = 0; k < c->M; k++) {
    for (int j = 0; j < 3; j++) {
      double d = fabs(a->vecs[k][j] - (b ? b->vecs[k][j] : 0.0));
      acc = inf_norm ? (d > acc ? d : acc) : acc + d * d;
    }
    if (c->colmap[3 * k + 2] >= 0) {
      double d = fabs(a->t[k] - (b ? b->t[k] : 0.0));
      acc = inf_norm ? (d > acc ? d : acc) : acc + d * d;
    }
  }
  return inf_norm ? acc : sqrt(acc);
}

typedef struct {
  int max_consecutive_nonmonotonic_steps;
  double minimum_cost, current_cost, reference_cost, candidate_cost;
  double accumulated_reference_model_cost_change, accumulated_candidate_model_cost_change;
  int num_consecutive_nonmonotonic_steps;
} step_eval_t;

static void se_init(step_eval_t *e, double initial_cost, int max_nonmono) {
  e->max_consecutive_nonmonotonic_steps = max_nonmono;
  e->minimum_cost = e->current_cost = e->reference_cost = e->candidate_cost = initial_cost;
  e->accumulated_reference_model_cost_change = 0.0;
  e->accumulated_candidate_model_cost_change = 0.0;
  e->num_consecutive_nonmonotonic_steps = 0;
}
static double se_quality(const step_eval_t *e, double cost, double model_cost_change) {
  double rel = (e->current_cost - cost) / model_cost_change;
  double hist = (e->reference_cost - cost) /
                (e->accumulated_reference_model_cost_change + model_cost_change);
  return rel > hist ? rel : hist;
}
static void se_accept(step_eval_t *e, double cost, double model_cost_change) {
  e->current_cost = cost;
  e->accumulated_candidate_model_cost_change += model_cost_change;
  e->accumulated_reference_model_cost_change += model_cost_change;
  if (e->current_cost < e->minimum_cost) {
    e->minimum_cost = e->current_cost;
    e->num_consecutive_nonmonotonic_steps = 0;
    e->candidate_cost = e->current_cost;
    e->accumulated_candidate_model_cost_change = 0.0;
  } else {
    ++e->num_consecutive_nonmonotonic_steps;
    if (e->current_cost > e->candidate_cost) {
      e->candidate_cost = e->current_cost;
      e->accumulated_candidate_model_cost_change = 0.0;
    }
  }
  if (e->num_consecutive_nonmonotonic_steps == e->max_consecutive_nonmonotonic_steps) {
    e->reference_cost = e->candidate_cost;
    e->accumulated_reference_model_cost_change = e->accumulated_candidate_model_cost_change;
  }
}

int32_t mvo_oracle_refine(mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                          const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                          const double *m0, const double *m1, mvo_oracle_summary *summary,
                          mvo_oracle_iteration *trace, int32_t trace_capacity,
                          int32_t *trace_len) {
  return mvo_oracle_refine_from(frames, M, N, opt, pair_offsets, m0, m1, NULL, summary, trace, trace_capacity,
                                trace_len);
}

int32_t mvo_oracle_refine_from(mvo_oracle_frame *frames, uint32_t M, uint32_t N,
                               const mvo_oracle_options *opt, const uint64_t *pair_offsets,
                               const double *m0, const double *m1, const double *point,
                               mvo_oracle_summary *summary, mvo_oracle_iteration *trace,
                               int32_t trace_capacity, int32_t *trace_len) {
  ctx_t c;
  if (ctx_init(&c, frames, M, N, opt, pair_offsets, m0, m1)) return -1;
  const int n = c.ncol;
  const uint64_t m = c.n_obs;

  /* Ceres Solver::Options defaults not overridden at IR.cpp:63-76 */
  const double min_relative_decrease = 1e-3;
  const double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  const double max_radius = 1e16, min_radius = 1e-32;
  const int max_consecutive_nonmonotonic_steps = 5; /* use_nonmonotonic_steps = true, :72 */
  const int max_consecutive_invalid_steps = 5;
  double radius = 1e4, decrease_factor = 2.0;
  int reuse_diagonal = 0;

  double *J = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1) * n);
  double *res = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1));
  if (!J || !res) {
    free(J);
    free(res);
    return -2;
  }
  double grad[MAXC], jscale_[MAXC], diag[MAXC], lmd[MAXC], step[MAXC], delta[MAXC];
  point_t x, cand, best, proj, start;
  initial_point(&c, x.vecs, x.t);
  if (point) /* test aid: start somewhere reverseScale cannot reach (t = -800: the saturated branch) */
    for (uint32_t k = 0; k < M; k++) {
      for (int j = 0; j < 3; j++) x.vecs[k][j] = point[4 * k + j];
      x.t[k] = point[4 * k + 3];
    }
  best = x;
  start = x;

  mvo_oracle_summary S;
  memset(&S, 0, sizeof S);
  S.termination = MVO_ORACLE_NO_CONVERGENCE;
  int32_t ntrace = 0;
  double minimum_cost = DBL_MAX, x_cost, x_norm, model_cost_change = 0.0;
  double final_cost;

  mvo_oracle_iteration it;
  memset(&it, 0, sizeof it);
  int iteration = 0, num_invalid = 0;

  /* ---- IterationZero + EvaluateGradientAndJacobian ---- */
  x_norm = point_diff_norm(&c, &x, NULL, 0);
  x_cost = evaluate(&c, (const double(*)[3])x.vecs, x.t, 0, res, J, grad);
  if (!isfinite(x_cost)) {
    /* Ceres 1.14: ResidualBlock::Evaluate rejects non-finite residuals (IsEvaluationValid), so
     * TrustRegionMinimizer::IterationZero fails: "Residual and Jacobian evaluation failed." =>
     * FAILURE before any step; the reference logs it and writes the untouched parameters back
     * (IR.cpp:235-246) */
    S.termination = MVO_ORACLE_FAILURE;
    S.reason = MVO_ORACLE_REASON_EVALUATION_FAILED;
    S.initial_cost = S.final_cost = x_cost;
    for (uint32_t k = 0; k < M; k++) {
      frames[k].scale = mvo_oracle_scale(x.t[k], opt->highest_length, opt->lowest_length);
      for (int j = 0; j < 3; j++) frames[k].vec[j] = x.vecs[k][j];
    }
    if (summary) *summary = S;
    if (trace_len) *trace_len = 0;
    free(J);
    free(res);
    return 0;
  }
  for (int j = 0; j < n; j++) {
    double s2 = 0.0;
    for (uint64_t i = 0; i < m; i++) s2 += J[i * n + j] * J[i * n + j];
    jscale_[j] = 1.0 / (1.0 + sqrt(s2)); /* jacobi_scaling, fixed at iteration 0 */
  }
  for (uint64_t i = 0; i < m; i++)
    for (int j = 0; j < n; j++) J[i * n + j] *= jscale_[j];
  {
    double ng[MAXC];
    for (int j = 0; j < n; j++) ng[j] = -grad[j];
    point_plus(&c, &x, ng, &proj);
    it.gradient_max_norm = point_diff_norm(&c, &x, &proj, 1);
  }
  it.cost = x_cost;
  it.step_is_valid = 1;
  it.step_is_successful = 1;
  S.initial_cost = x_cost;
  final_cost = x_cost;
  double prev_gradient_max_norm = it.gradient_max_norm;

  step_eval_t se;
  se_init(&se, x_cost, max_consecutive_nonmonotonic_steps);

  for (;;) {
    /* ---- FinalizeIterationAndCheckIfMinimizerCanContinue ---- */
    if (it.step_is_successful) {
      ++S.num_successful_steps;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
        best = x;
      }
    } else {
      ++S.num_unsuccessful_steps;
    }
    it.trust_region_radius = radius;
    if (trace && ntrace < trace_capacity) trace[ntrace] = it;
    ++ntrace;
    if (it.cost < final_cost) final_cost = it.cost;
    S.iterations = iteration;
    prev_gradient_max_norm = it.gradient_max_norm;

    if (iteration >= opt->max_num_iterations) {
      S.termination = MVO_ORACLE_NO_CONVERGENCE;
      S.reason = MVO_ORACLE_REASON_MAX_ITERATIONS;
      break;
    }
    if (it.step_is_successful && it.gradient_max_norm <= opt->gradient_tolerance) {
      S.termination = MVO_ORACLE_CONVERGENCE;
      S.reason = MVO_ORACLE_REASON_GRADIENT_TOLERANCE;
      break;
    }
    if (radius <= min_radius) {
      S.termination = MVO_ORACLE_CONVERGENCE;
      S.reason = MVO_ORACLE_REASON_MIN_RADIUS;
      break;
    }

    memset(&it, 0, sizeof it);
    ++iteration;

    /* ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep) ---- */
    if (!reuse_diagonal) {
      for (int j = 0; j < n; j++) {
        double s2 = 0.0;
        for (uint64_t i = 0; i < m; i++) s2 += J[i * n + j] * J[i * n + j];
        diag[j] = fmin(fmax(s2, min_lm_diagonal), max_lm_diagonal);
      }
    }
    for (int j = 0; j < n; j++) lmd[j] = sqrt(diag[j] / radius);
    int lin_fail = lsq_qr(J, res, m, n, lmd, step);
    reuse_diagonal = 1;
    int valid = 0;
    if (!lin_fail) {
      for (int j = 0; j < n; j++) step[j] = -step[j];
      double mcc = 0.0;
      for (uint64_t i = 0; i < m; i++) {
        double mr = 0.0;
        for (int j = 0; j < n; j++) mr += J[i * n + j] * step[j];
        mcc += mr * (res[i] + mr / 2.0);
      }
      model_cost_change = -mcc;
      valid = model_cost_change > 0.0;
    }
    it.step_is_valid = valid;
    if (!valid) {
      /* ---- HandleInvalidStep ---- */
      if (++num_invalid >= max_consecutive_invalid_steps) {
        S.termination = MVO_ORACLE_FAILURE;
        S.reason = MVO_ORACLE_REASON_INVALID_STEPS;
        break;
      }
      radius *= 0.5; /* StepIsInvalid */
      reuse_diagonal = 1;
      it.cost = x_cost;
      it.cost_change = 0.0;
      it.gradient_max_norm = prev_gradient_max_norm;
      it.step_norm = 0.0;
      it.relative_decrease = 0.0;
      it.step_is_successful = 0;
      continue;
    }
    num_invalid = 0;
    for (int j = 0; j < n; j++) delta[j] = step[j] * jscale_[j];

    /* ---- ComputeCandidatePointAndEvaluateCost ---- */
    point_plus(&c, &x, delta, &cand);
    double cand_cost = evaluate(&c, (const double(*)[3])cand.vecs, cand.t, 0, NULL, NULL, NULL);
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX; /* Ceres: "Step failed to evaluate" => candidate_cost_ = max */

    /* ---- ParameterToleranceReached ---- */
    it.step_norm = point_diff_norm(&c, &x, &cand, 0);
    if (it.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      S.termination = MVO_ORACLE_CONVERGENCE;
      S.reason = MVO_ORACLE_REASON_PARAMETER_TOLERANCE;
      break;
    }
    /* ---- FunctionToleranceReached ---- */
    it.cost_change = x_cost - cand_cost;
    if (fabs(it.cost_change) <= opt->function_tolerance * x_cost) {
      S.termination = MVO_ORACLE_CONVERGENCE;
      S.reason = MVO_ORACLE_REASON_FUNCTION_TOLERANCE;
      break;
    }
    /* ---- IsStepSuccessful ---- */
    it.relative_decrease = se_quality(&se, cand_cost, model_cost_change);
    if (it.relative_decrease > min_relative_decrease) {
      /* ---- HandleSuccessfulStep ---- */
      x = cand;
      x_norm = point_diff_norm(&c, &x, NULL, 0);
      x_cost = evaluate(&c, (const double(*)[3])x.vecs, x.t, 0, res, J, grad);
      for (uint64_t i = 0; i < m; i++)
        for (int j = 0; j < n; j++) J[i * n + j] *= jscale_[j];
      double ng[MAXC];
      for (int j = 0; j < n; j++) ng[j] = -grad[j];
      point_plus(&c, &x, ng, &proj);
      it.gradient_max_norm = point_diff_norm(&c, &x, &proj, 1);
      it.cost = x_cost;
      it.step_is_successful = 1;
      /* LevenbergMarquardtStrategy::StepAccepted */
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = fmin(max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
      se_accept(&se, cand_cost, model_cost_change);
    } else {
      /* ---- HandleUnsuccessfulStep ---- */
      it.step_is_successful = 0;
      radius = radius / decrease_factor; /* StepRejected */
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      it.cost = cand_cost;
      it.gradient_max_norm = prev_gradient_max_norm;
    }
  }

  S.final_cost = final_cost;
  /* IR.cpp:240-246: write back.  Ceres leaves the lowest-cost iterate in the user's arrays when the
   * summary is usable (Solver::Summary::IsSolutionUsable: CONVERGENCE / NO_CONVERGENCE); after
   * FAILURE it restores the INITIAL parameter values, so the reference writes
   * s(reverseScale(scale_0)) and vec_0 back. */
  if (S.termination == MVO_ORACLE_FAILURE) best = start;
  for (uint32_t k = 0; k < M; k++) {
    frames[k].scale = mvo_oracle_scale(best.t[k], opt->highest_length, opt->lowest_length);
    for (int j = 0; j < 3; j++) frames[k].vec[j] = best.vecs[k][j];
  }
  if (summary) *summary = S;
  if (trace_len) *trace_len = ntrace;
  free(J);
  free(res);
  return 0;
}

/* mvo_ba_oracle.c -- see mvo_ba_oracle.h.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the
 * reference has no explicit-landmark bundle adjustment; this is a self-oracle (dense, scalar). */
#include "mvo_ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void loss_eval(const mvo_ba_options *opt, double s, double *rho, double *w) {
  const double a = opt->loss_scale, b = a * a;
  if (opt->loss == 1) { /* Cauchy */
    *rho = b * log(1.0 + s / b);
    *w = 1.0 / (1.0 + s / b);
  } else if (opt->loss == 2 && s > b) { /* Huber, outlier branch */
    const double rt = sqrt(s);
    *rho = 2.0 * a * rt - b;
    *w = a / rt;
  } else {
    *rho = s;
    *w = 1.0;
  }
}

void mvo_ba_oracle_obs_jacobian(const mvo_ba_pose *pose, const double *X, const double *ray, double r[2],
                                double jc[12], double jx[6]) {
  const double *R = pose->R;
  const double d[3] = {X[0] - pose->p[0], X[1] - pose->p[1], X[2] - pose->p[2]};
  /* q = R^T d */
  const double a = R[0] * d[0] + R[3] * d[1] + R[6] * d[2];
  const double b = R[1] * d[0] + R[4] * d[1] + R[7] * d[2];
  const double c = R[2] * d[0] + R[5] * d[1] + R[8] * d[2];
  const double ic = 1.0 / c;
  r[0] = a * ic - ray[0] / ray[2];
  r[1] = b * ic - ray[1] / ray[2];
  /* Pi = [[1/c, 0, -a/c^2], [0, 1/c, -b/c^2]];  J_X = Pi R^T */
  const double pa = -a * ic * ic, pb = -b * ic * ic;
  for (int k = 0; k < 3; ++k) {
    /* column k of R^T-rows: (R^T)[0][k] = R[k][0] ... */
    jx[k] = ic * R[3 * k + 0] + pa * R[3 * k + 2];
    jx[3 + k] = ic * R[3 * k + 1] + pb * R[3 * k + 2];
  }
  /* J_omega = Pi [q]x */
  jc[0] = a * b * ic * ic;
  jc[1] = -1.0 - a * a * ic * ic;
  jc[2] = b * ic;
  jc[6] = 1.0 + b * b * ic * ic;
  jc[7] = -a * b * ic * ic;
  jc[8] = -a * ic;
  /* J_p = -J_X */
  for (int k = 0; k < 3; ++k) {
    jc[3 + k] = -jx[k];
    jc[9 + k] = -jx[3 + k];
  }
}

void mvo_ba_oracle_pose_plus(const mvo_ba_pose *in, const double delta[6], mvo_ba_pose *out) {
  const double wx = delta[0], wy = delta[1], wz = delta[2];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double A, B; /* Exp(w) = I + A [w]x + B [w]x^2 */
  if (th < 1e-8) {
    A = 1.0 - th2 / 6.0;
    B = 0.5 - th2 / 24.0;
  } else {
    A = sin(th) / th;
    B = (1.0 - cos(th)) / th2;
  }
  const double E[9] = {1.0 - B * (wy * wy + wz * wz), -A * wz + B * wx * wy, A * wy + B * wx * wz,
                       A * wz + B * wx * wy, 1.0 - B * (wx * wx + wz * wz), -A * wx + B * wy * wz,
                       -A * wy + B * wx * wz, A * wx + B * wy * wz, 1.0 - B * (wx * wx + wy * wy)};
  double Rn[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Rn[3 * i + j] = in->R[3 * i] * E[j] + in->R[3 * i + 1] * E[3 + j] + in->R[3 * i + 2] * E[6 + j];
  memcpy(out->R, Rn, sizeof Rn);
  for (int k = 0; k < 3; ++k) out->p[k] = in->p[k] + delta[3 + k];
}

void mvo_ba_oracle_residuals(uint32_t N, const mvo_ba_pose *poses, uint32_t L, const double *X,
                             const uint64_t *off, const uint32_t *obs_frame, const double *obs_ray,
                             const mvo_ba_options *opt, double *residuals, double *cost) {
  (void)N;
  double c = 0.0;
  for (uint32_t j = 0; j < L; ++j)
    for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
      double r[2], jc[12], jx[6], rho, w;
      mvo_ba_oracle_obs_jacobian(&poses[obs_frame[o]], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
      if (residuals) {
        residuals[2 * o] = r[0];
        residuals[2 * o + 1] = r[1];
      }
      loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
      c += 0.5 * rho;
    }
  if (cost) *cost = c;
}

static void inv3_sym(const double C[9], double Ci[9]) {
  const double a = C[0], b = C[1], c = C[2], d = C[4], e = C[5], f = C[8];
  const double co0 = d * f - e * e, co1 = c * e - b * f, co2 = b * e - c * d;
  const double det = a * co0 + b * co1 + c * co2, id = 1.0 / det;
  Ci[0] = co0 * id;
  Ci[1] = Ci[3] = co1 * id;
  Ci[2] = Ci[6] = co2 * id;
  Ci[4] = (a * f - c * c) * id;
  Ci[5] = Ci[7] = (b * c - a * e) * id;
  Ci[8] = (a * d - b * b) * id;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Everything of one linearisation.  Returns per-landmark data needed for the back substitution in
 * Cinv (9 L), gl (3 L), Dl2 (3 L) when those are non-NULL. */
static void reduced_system(uint32_t N, uint32_t M, const mvo_ba_pose *poses, uint32_t L, const double *X,
                           const uint64_t *off, const uint32_t *obs_frame, const double *obs_ray,
                           const mvo_ba_options *opt, double radius, double *S, double *rhs, double *Dc2,
                           double *Cinv_out, double *gl_out, double *Dl2_out, double *gmax, double *cost) {
  (void)N;
  const uint32_t n = 6 * M;
  memset(S, 0, sizeof(double) * n * n);
  memset(rhs, 0, sizeof(double) * n);
  double c = 0.0, gm = 0.0;
  /* pass 1: B (into S), g_c (into rhs) */
  for (uint32_t j = 0; j < L; ++j)
    for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
      const uint32_t f = obs_frame[o];
      double r[2], jc[12], jx[6], rho, w;
      mvo_ba_oracle_obs_jacobian(&poses[f], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
      loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
      c += 0.5 * rho;
      if (f >= M) continue;
      for (int p = 0; p < 6; ++p) {
        rhs[6 * f + p] += w * (jc[p] * r[0] + jc[6 + p] * r[1]);
        for (int q = 0; q < 6; ++q) S[(6 * f + p) * n + 6 * f + q] += w * (jc[p] * jc[q] + jc[6 + p] * jc[6 + q]);
      }
    }
  for (uint32_t i = 0; i < n; ++i) {
    Dc2[i] = clampd(S[i * n + i], 1e-6, 1e32) / radius;
    if (fabs(rhs[i]) > gm) gm = fabs(rhs[i]);
  }
  for (uint32_t i = 0; i < n; ++i) S[i * n + i] += Dc2[i];
  /* pass 2: landmarks and the Schur complement */
  for (uint32_t j = 0; j < L; ++j) {
    double C[9] = {0}, gl[3] = {0, 0, 0};
    for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
      double r[2], jc[12], jx[6], rho, w;
      mvo_ba_oracle_obs_jacobian(&poses[obs_frame[o]], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
      loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
      for (int p = 0; p < 3; ++p) {
        gl[p] += w * (jx[p] * r[0] + jx[3 + p] * r[1]);
        for (int q = 0; q < 3; ++q) C[3 * p + q] += w * (jx[p] * jx[q] + jx[3 + p] * jx[3 + q]);
      }
    }
    double Dl2[3], Ci[9];
    for (int p = 0; p < 3; ++p) {
      if (fabs(gl[p]) > gm) gm = fabs(gl[p]);
      Dl2[p] = clampd(C[4 * p], 1e-6, 1e32) / radius;
      C[4 * p] += Dl2[p];
    }
    inv3_sym(C, Ci);
    if (Cinv_out) memcpy(Cinv_out + 9 * j, Ci, sizeof Ci);
    if (gl_out) memcpy(gl_out + 3 * j, gl, sizeof gl);
    if (Dl2_out) memcpy(Dl2_out + 3 * j, Dl2, sizeof Dl2);
    /* Y_o = E_o C^-1 for every observation in a free frame; S -= Y_o E_o'^T; rhs -= Y_o g_l */
    for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
      const uint32_t f = obs_frame[o];
      if (f >= M) continue;
      double r[2], jc[12], jx[6], rho, w, E[18], Y[18];
      mvo_ba_oracle_obs_jacobian(&poses[f], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
      loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
      for (int p = 0; p < 6; ++p)
        for (int q = 0; q < 3; ++q) E[3 * p + q] = w * (jc[p] * jx[q] + jc[6 + p] * jx[3 + q]);
      for (int p = 0; p < 6; ++p)
        for (int q = 0; q < 3; ++q) Y[3 * p + q] = E[3 * p] * Ci[q] + E[3 * p + 1] * Ci[3 + q] + E[3 * p + 2] * Ci[6 + q];
      for (int p = 0; p < 6; ++p) rhs[6 * f + p] -= Y[3 * p] * gl[0] + Y[3 * p + 1] * gl[1] + Y[3 * p + 2] * gl[2];
      for (uint64_t o2 = off[j]; o2 < off[j + 1]; ++o2) {
        const uint32_t f2 = obs_frame[o2];
        if (f2 >= M) continue;
        double r2[2], jc2[12], jx2[6], rho2, w2;
        mvo_ba_oracle_obs_jacobian(&poses[f2], X + 3 * j, obs_ray + 3 * o2, r2, jc2, jx2);
        loss_eval(opt, r2[0] * r2[0] + r2[1] * r2[1], &rho2, &w2);
        for (int p = 0; p < 6; ++p)
          for (int q = 0; q < 6; ++q) {
            double acc = 0.0;
            for (int t = 0; t < 3; ++t) acc += Y[3 * p + t] * (w2 * (jc2[q] * jx2[t] + jc2[6 + q] * jx2[3 + t]));
            S[(6 * f + p) * n + 6 * f2 + q] -= acc;
          }
      }
    }
  }
  if (gmax) *gmax = gm;
  if (cost) *cost = c;
}

void mvo_ba_oracle_reduced(uint32_t N, uint32_t M, const mvo_ba_pose *poses, uint32_t L, const double *X,
                           const uint64_t *off, const uint32_t *obs_frame, const double *obs_ray,
                           const mvo_ba_options *opt, double radius, double *S, double *rhs, double *gradient,
                           double *cost) {
  const uint32_t n = 6 * M;
  double *Dc2 = (double *)malloc(sizeof(double) * (n + 1));
  double *gl = (double *)malloc(sizeof(double) * 3 * (L + 1));
  /* rhs holds the reduced right-hand side; the raw camera gradient is recomputed for `gradient` */
  reduced_system(N, M, poses, L, X, off, obs_frame, obs_ray, opt, radius, S, rhs, Dc2, NULL, gl, NULL, NULL, cost);
  if (gradient) {
    memset(gradient, 0, sizeof(double) * n);
    for (uint32_t j = 0; j < L; ++j)
      for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
        const uint32_t f = obs_frame[o];
        if (f >= M) continue;
        double r[2], jc[12], jx[6], rho, w;
        mvo_ba_oracle_obs_jacobian(&poses[f], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
        loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
        for (int p = 0; p < 6; ++p) gradient[6 * f + p] += w * (jc[p] * r[0] + jc[6 + p] * r[1]);
      }
    memcpy(gradient + n, gl, sizeof(double) * 3 * L);
  }
  free(Dc2);
  free(gl);
}

/* dense Cholesky solve of S x = b (S symmetric positive definite, n x n); returns 0 on success */
static int chol_solve(uint32_t n, double *S, double *b) {
  for (uint32_t k = 0; k < n; ++k) {
    double d = S[k * n + k];
    for (uint32_t t = 0; t < k; ++t) d -= S[k * n + t] * S[k * n + t];
    if (!(d > 0.0) || isinf(d)) return 1;
    d = sqrt(d);
    S[k * n + k] = d;
    for (uint32_t i = k + 1; i < n; ++i) {
      double v = S[i * n + k];
      for (uint32_t t = 0; t < k; ++t) v -= S[i * n + t] * S[k * n + t];
      S[i * n + k] = v / d;
    }
  }
  for (uint32_t i = 0; i < n; ++i) {
    double v = b[i];
    for (uint32_t t = 0; t < i; ++t) v -= S[i * n + t] * b[t];
    b[i] = v / S[i * n + i];
  }
  for (uint32_t ii = n; ii-- > 0;) {
    double v = b[ii];
    for (uint32_t t = ii + 1; t < n; ++t) v -= S[t * n + ii] * b[t];
    b[ii] = v / S[ii * n + ii];
  }
  return 0;
}

int mvo_ba_oracle_refine(uint32_t N, uint32_t M, mvo_ba_pose *poses, uint32_t L, double *X,
                         const uint64_t *off, const uint32_t *obs_frame, const double *obs_ray,
                         const mvo_ba_options *opt, mvo_ba_summary *summary, mvo_ba_iteration *trace,
                         int32_t trace_capacity, int32_t *trace_rows) {
  const uint32_t n = 6 * M;
  double *S = (double *)malloc(sizeof(double) * (n * n + 1));
  double *rhs = (double *)malloc(sizeof(double) * (n + 1));
  double *gc = (double *)malloc(sizeof(double) * (n + 1));
  double *Dc2 = (double *)malloc(sizeof(double) * (n + 1));
  double *Cinv = (double *)malloc(sizeof(double) * 9 * (L + 1));
  double *gl = (double *)malloc(sizeof(double) * 3 * (L + 1));
  double *Dl2 = (double *)malloc(sizeof(double) * 3 * (L + 1));
  double *Xc = (double *)malloc(sizeof(double) * 3 * (L + 1));
  mvo_ba_pose *pc = (mvo_ba_pose *)malloc(sizeof(mvo_ba_pose) * N);
  double radius = opt->initial_radius > 0.0 ? opt->initial_radius : 1e4, decf = 2.0;
  double cost = 0.0;
  int rows = 0, nsucc = 0, nunsucc = 0, num_invalid = 0, termination = 1, iteration = 0;
  mvo_ba_oracle_residuals(N, poses, L, X, off, obs_frame, obs_ray, opt, NULL, &cost);
  const double initial_cost = cost;
  {
    mvo_ba_iteration it;
    memset(&it, 0, sizeof it);
    it.cost = cost;
    it.trust_region_radius = radius;
    it.step_is_valid = it.step_is_successful = 1;
    if (trace && rows < trace_capacity) trace[rows] = it;
    ++rows;
  }
  while (iteration < opt->max_num_iterations) {
    ++iteration;
    mvo_ba_iteration it;
    memset(&it, 0, sizeof it);
    double gmax = 0.0, cost_lin = 0.0;
    reduced_system(N, M, poses, L, X, off, obs_frame, obs_ray, opt, radius, S, rhs, Dc2, Cinv, gl, Dl2, &gmax,
                   &cost_lin);
    it.gradient_max_norm = gmax;
    if (gmax <= opt->gradient_tolerance) {
      termination = 0;
      --iteration;
      break;
    }
    /* raw camera gradient g_c = rhs + sum_j Y g_l: recompute directly for the model cost change */
    memset(gc, 0, sizeof(double) * n);
    for (uint32_t j = 0; j < L; ++j)
      for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
        const uint32_t f = obs_frame[o];
        if (f >= M) continue;
        double r[2], jc[12], jx[6], rho, w;
        mvo_ba_oracle_obs_jacobian(&poses[f], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
        loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
        for (int p = 0; p < 6; ++p) gc[6 * f + p] += w * (jc[p] * r[0] + jc[6 + p] * r[1]);
      }
    int bad = chol_solve(n, S, rhs); /* rhs <- S^-1 rhs; dc = -that */
    double mcc = 0.0, step2 = 0.0, xnorm2 = 0.0;
    if (!bad) {
      for (uint32_t i = 0; i < n; ++i) {
        const double dc = -rhs[i];
        rhs[i] = dc;
        if (!isfinite(dc)) bad = 1;
        mcc += dc * (Dc2[i] * dc - gc[i]);
        step2 += dc * dc;
      }
    }
    if (!bad) {
      memcpy(pc, poses, sizeof(mvo_ba_pose) * N);
      for (uint32_t f = 0; f < M; ++f) {
        mvo_ba_oracle_pose_plus(&poses[f], rhs + 6 * f, &pc[f]);
        for (int k = 0; k < 3; ++k) xnorm2 += poses[f].p[k] * poses[f].p[k];
      }
      /* dl_j = -C^-1 (g_l + E^T dc) */
      for (uint32_t j = 0; j < L; ++j) {
        double t[3] = {gl[3 * j], gl[3 * j + 1], gl[3 * j + 2]};
        for (uint64_t o = off[j]; o < off[j + 1]; ++o) {
          const uint32_t f = obs_frame[o];
          if (f >= M) continue;
          double r[2], jc[12], jx[6], rho, w;
          mvo_ba_oracle_obs_jacobian(&poses[f], X + 3 * j, obs_ray + 3 * o, r, jc, jx);
          loss_eval(opt, r[0] * r[0] + r[1] * r[1], &rho, &w);
          double jd[2] = {0, 0}; /* J_c dc */
          for (int p = 0; p < 6; ++p) {
            jd[0] += jc[p] * rhs[6 * f + p];
            jd[1] += jc[6 + p] * rhs[6 * f + p];
          }
          for (int q = 0; q < 3; ++q) t[q] += w * (jx[q] * jd[0] + jx[3 + q] * jd[1]);
        }
        const double *Ci = Cinv + 9 * j;
        for (int q = 0; q < 3; ++q) {
          const double dl = -(Ci[3 * q] * t[0] + Ci[3 * q + 1] * t[1] + Ci[3 * q + 2] * t[2]);
          if (!isfinite(dl)) bad = 1;
          Xc[3 * j + q] = X[3 * j + q] + dl;
          mcc += dl * (Dl2[3 * j + q] * dl - gl[3 * j + q]);
          step2 += dl * dl;
          xnorm2 += X[3 * j + q] * X[3 * j + q];
        }
      }
      mcc *= 0.5;
    }
    if (bad || !(mcc > 0.0)) { /* invalid step */
      if (++num_invalid >= 5) {
        termination = 2;
        break;
      }
      radius *= 0.5;
      it.cost = cost;
      it.trust_region_radius = radius;
      ++nunsucc;
      if (trace && rows < trace_capacity) trace[rows] = it;
      ++rows;
      continue;
    }
    num_invalid = 0;
    it.step_is_valid = 1;
    it.step_norm = sqrt(step2);
    if (it.step_norm <= opt->parameter_tolerance * (sqrt(xnorm2) + opt->parameter_tolerance)) {
      termination = 0;
      --iteration;
      break;
    }
    double cost_c = 0.0;
    mvo_ba_oracle_residuals(N, pc, L, Xc, off, obs_frame, obs_ray, opt, NULL, &cost_c);
    it.cost_change = cost - cost_c;
    it.relative_decrease = it.cost_change / mcc;
    const double cost_before = cost;
    if (it.relative_decrease > 1e-3) {
      memcpy(poses, pc, sizeof(mvo_ba_pose) * M);
      memcpy(X, Xc, sizeof(double) * 3 * L);
      cost = cost_c;
      const double q = 2.0 * it.relative_decrease - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
      radius = fmin(1e16, radius);
      decf = 2.0;
      it.step_is_successful = 1;
      ++nsucc;
    } else {
      radius = radius / decf;
      decf *= 2.0;
      ++nunsucc;
    }
    it.cost = cost;
    it.trust_region_radius = radius;
    if (trace && rows < trace_capacity) trace[rows] = it;
    ++rows;
    if (it.step_is_successful && fabs(it.cost_change) <= opt->function_tolerance * cost_before) {
      termination = 0;
      break;
    }
    if (radius <= 1e-32) {
      termination = 0;
      break;
    }
  }
  if (summary) {
    summary->termination = termination;
    summary->iterations = iteration;
    summary->num_successful_steps = nsucc;
    summary->num_unsuccessful_steps = nunsucc;
    summary->initial_cost = initial_cost;
    summary->final_cost = cost;
  }
  if (trace_rows) *trace_rows = rows;
  free(S); free(rhs); free(gc); free(Dc2); free(Cinv); free(gl); free(Dl2); free(Xc); free(pc);
  return 0;
}

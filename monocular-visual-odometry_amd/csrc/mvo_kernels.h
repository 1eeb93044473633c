// mvo_kernels.h -- hand-written gfx950 kernels of libmvo_hip (included once by mvo_hip.hip).
//
//   k_fold   observation pass #0: c = (R_a m0) x (R_{b+1} m1) per residual block, optional
//            store of c (SoA planes) and/or per-tile second moments S = sum c c^T.
//   k_pass   evaluation pass: r = u_p . c, Cauchy weight, weighted moments + sum rho.
//   (k_lm, the device-resident Levenberg-Marquardt loop, lives in mvo_lm.h)
//   k_eval   debug: per-residual-block r and local Jacobian row.
//
// Math: r_i = u_p . c_i with the per-pair unit baseline u_p, so the
// Jacobian row of block i is c_i^T A_p with a per-pair 3 x 3|F_p| matrix A_p, and
//   J^T J = sum_p A_p^T S_p A_p,  J^T r = sum_p A_p^T S_p u_p,  S_p = sum_i w_i c_i c_i^T.
// All reductions are wavefront (64-lane) shuffles -> LDS -> one partial per workgroup slot;
// no global atomics, fixed summation order (bitwise reproducible).
#pragma once

#include "mvo_device.h"

#define FOLD_THREADS 256
#define FOLD_WAVES 4

// ---------------------------------------------------------------------------------------
// workgroup epilogue shared by k_fold / k_pass: red[w][8] holds each wave's tile sums; the
// waves of a workgroup whose tiles share a slot are summed in wave order and stored once.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void block_store_partials(const TileDev* __restrict__ tiles, uint32_t n_tiles,
                                                     double (*red)[8], double* __restrict__ partial) {
  __syncthreads();
  const int v = threadIdx.x;
  if (v < 8) {
    double acc = 0.0;
    const uint32_t t0 = blockIdx.x * FOLD_WAVES;
#pragma unroll
    for (int w = 0; w < FOLD_WAVES; ++w) {
      const uint32_t t = t0 + w;
      if (t >= n_tiles) break;
      acc += red[w][v];
      const uint32_t slot = tiles[t].slot;
      const bool last = (w == FOLD_WAVES - 1) || (t + 1 >= n_tiles) || (tiles[t + 1].slot != slot);
      if (last) {
        partial[(size_t)slot * 8 + v] = acc;
        acc = 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_fold: reads m0, m1 (48 B / residual block, caller's interleaved layout), optionally writes
// c (24 B / block) and/or accumulates S.  One wavefront per tile.
// ---------------------------------------------------------------------------------------
template <bool WRITE_C, bool MOMENTS>
__global__ __launch_bounds__(FOLD_THREADS) void k_fold(const TileDev* __restrict__ tiles, uint32_t n_tiles,
                                                       const double* __restrict__ frames,
                                                       const double* __restrict__ m0,
                                                       const double* __restrict__ m1,
                                                       double* __restrict__ cbuf, size_t cstride,
                                                       double* __restrict__ partial) {
  __shared__ double red[FOLD_WAVES][8];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * FOLD_WAVES + wv;
  double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0;
  if (t < n_tiles) {
    const TileDev tl = tiles[t];
    const double* __restrict__ R0 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f0;
    const double* __restrict__ R1 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f1;
    const double a0 = R0[0], a1 = R0[1], a2 = R0[2], a3 = R0[3], a4 = R0[4], a5 = R0[5], a6 = R0[6],
                 a7 = R0[7], a8 = R0[8];
    const double b0 = R1[0], b1 = R1[1], b2 = R1[2], b3 = R1[3], b4 = R1[4], b5 = R1[5], b6 = R1[6],
                 b7 = R1[7], b8 = R1[8];
    const double* __restrict__ p0 = m0 + 3 * (size_t)tl.src_start;
    const double* __restrict__ p1 = m1 + 3 * (size_t)tl.src_start;
    double* __restrict__ cx = cbuf + tl.c_start;
    double* __restrict__ cy = cx + cstride;
    double* __restrict__ cz = cy + cstride;
#pragma unroll 4
    for (uint32_t i = lane; i < tl.count; i += 64) {
      const double x0 = p0[3 * i], y0 = p0[3 * i + 1], z0 = p0[3 * i + 2];
      const double x1 = p1[3 * i], y1 = p1[3 * i + 1], z1 = p1[3 * i + 2];
      const double px = a0 * x0 + a1 * y0 + a2 * z0;  // R0 * m0
      const double py = a3 * x0 + a4 * y0 + a5 * z0;
      const double pz = a6 * x0 + a7 * y0 + a8 * z0;
      const double qx = b0 * x1 + b1 * y1 + b2 * z1;  // R1 * m1
      const double qy = b3 * x1 + b4 * y1 + b5 * z1;
      const double qz = b6 * x1 + b7 * y1 + b8 * z1;
      const double vx = py * qz - pz * qy;  // c = p x q
      const double vy = pz * qx - px * qz;
      const double vz = px * qy - py * qx;
      if (WRITE_C) {
        cx[i] = vx;
        cy[i] = vy;
        cz[i] = vz;
      }
      if (MOMENTS) {
        S0 += vx * vx;
        S1 += vx * vy;
        S2 += vx * vz;
        S3 += vy * vy;
        S4 += vy * vz;
        S5 += vz * vz;
      }
    }
  }
  if (MOMENTS) {
    S0 = wave_sum(S0);
    S1 = wave_sum(S1);
    S2 = wave_sum(S2);
    S3 = wave_sum(S3);
    S4 = wave_sum(S4);
    S5 = wave_sum(S5);
    if (lane == 0) {
      red[wv][0] = S0;
      red[wv][1] = S1;
      red[wv][2] = S2;
      red[wv][3] = S3;
      red[wv][4] = S4;
      red[wv][5] = S5;
      red[wv][6] = 0.0;
      red[wv][7] = 0.0;
    }
    block_store_partials(tiles, n_tiles, red, partial);
  }
}

// ---------------------------------------------------------------------------------------
// k_pass: one evaluation of all residual blocks at the unit baselines in pair_u.
// reads c (24 B / block) as 16-byte loads from three SoA planes.
// partial[slot] = { S (6, weighted), sum rho(r^2) or sum r^2, sum w r^2 }
// ---------------------------------------------------------------------------------------
template <bool ROBUST>
__device__ __forceinline__ void pass_accum(double cx, double cy, double cz, double u0, double u1, double u2,
                                           double& S0, double& S1, double& S2, double& S3, double& S4,
                                           double& S5, double& rho, double& wr2) {
  const double r = u0 * cx + u1 * cy + u2 * cz;
  const double s = r * r;
  double w = 1.0;
  if (ROBUST) {
    // ceres::CauchyLoss(0.5): b = 0.25, c = 4; rho = b log(1 + s c); rho' = 1 / (1 + s c)
    const double sum = 1.0 + s * 4.0;
    const double inv = 1.0 / sum;
    rho += 0.25 * log(sum);
    w = fmax(DBL_MIN, inv);
  } else {
    rho += s;
  }
  wr2 += w * s;
  const double wx = w * cx, wy = w * cy, wz = w * cz;
  S0 += wx * cx;
  S1 += wx * cy;
  S2 += wx * cz;
  S3 += wy * cy;
  S4 += wy * cz;
  S5 += wz * cz;
}

template <bool ROBUST>
__global__ __launch_bounds__(FOLD_THREADS) void k_pass(const TileDev* __restrict__ tiles, uint32_t n_tiles,
                                                       const double* __restrict__ cbuf, size_t cstride,
                                                       const double* __restrict__ pair_u,
                                                       const int32_t* __restrict__ win_status,
                                                       double* __restrict__ partial) {
  __shared__ double red[FOLD_WAVES][8];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * FOLD_WAVES + wv;
  double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0, rho = 0, wr2 = 0;
  if (t < n_tiles) {
    const TileDev tl = tiles[t];
    if (win_status[tl.win] == 0) {
      const double u0 = pair_u[4 * (size_t)tl.pair], u1 = pair_u[4 * (size_t)tl.pair + 1],
                   u2 = pair_u[4 * (size_t)tl.pair + 2];
      const double* __restrict__ cx = cbuf + tl.c_start;
      const double* __restrict__ cy = cx + cstride;
      const double* __restrict__ cz = cy + cstride;
#pragma unroll 2
      for (uint32_t i = 2 * lane; i < tl.count; i += 128) {
        const double2 x = *reinterpret_cast<const double2*>(cx + i);
        const double2 y = *reinterpret_cast<const double2*>(cy + i);
        const double2 z = *reinterpret_cast<const double2*>(cz + i);
        pass_accum<ROBUST>(x.x, y.x, z.x, u0, u1, u2, S0, S1, S2, S3, S4, S5, rho, wr2);
        const bool v1 = (i + 1) < tl.count;  // the pad element of an odd tile is not data
        pass_accum<ROBUST>(v1 ? x.y : 0.0, v1 ? y.y : 0.0, v1 ? z.y : 0.0, u0, u1, u2, S0, S1, S2, S3,
                           S4, S5, rho, wr2);
      }
    }
  }
  S0 = wave_sum(S0);
  S1 = wave_sum(S1);
  S2 = wave_sum(S2);
  S3 = wave_sum(S3);
  S4 = wave_sum(S4);
  S5 = wave_sum(S5);
  rho = wave_sum(rho);
  wr2 = wave_sum(wr2);
  if (lane == 0) {
    red[wv][0] = S0;
    red[wv][1] = S1;
    red[wv][2] = S2;
    red[wv][3] = S3;
    red[wv][4] = S4;
    red[wv][5] = S5;
    red[wv][6] = rho;
    red[wv][7] = wr2;
  }
  block_store_partials(tiles, n_tiles, red, partial);
}

// ---------------------------------------------------------------------------------------
// k_reduce_slots: per-pair sums of the slot partials (observation-partitioned mode: these 8
// numbers per pair are what the ranks all-reduce before the identical LM step).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_slots(const PairDev* __restrict__ pairs, uint32_t n_pairs,
                                                      const double* __restrict__ partial,
                                                      double* __restrict__ pair_sums) {
  const uint32_t item = blockIdx.x * 256 + threadIdx.x;
  if (item >= n_pairs * 8) return;
  const uint32_t p = item >> 3, v = item & 7;
  const PairDev pd = pairs[p];
  double acc = 0.0;
  for (uint32_t sl = pd.slot_begin; sl < pd.slot_end; ++sl) acc += partial[(size_t)sl * 8 + v];
  pair_sums[item] = acc;
}

// ---------------------------------------------------------------------------------------
// k_eval (debug / parity): raw r and local Jacobian rows at the initial point.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(FOLD_THREADS) void k_eval(SolveParams prm, const TileDev* __restrict__ tiles,
                                                       uint32_t n_tiles, const WinDev* __restrict__ wins,
                                                       const PairDev* __restrict__ pairs,
                                                       const double* __restrict__ frames,
                                                       const double* __restrict__ init_x,
                                                       const double* __restrict__ m0, const double* __restrict__ m1,
                                                       double* __restrict__ res, double* __restrict__ jac) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * FOLD_WAVES + wv;
  if (t >= n_tiles) return;
  const TileDev tl = tiles[t];
  const WinDev win = wins[tl.win];
  const PairDev pd = pairs[tl.pair];
  const int M = win.M;
  const double* __restrict__ fr = frames + (size_t)MVO_FRAME_DOUBLES * win.frame_off;
  double zvec[MVO_MAXM][3], zt[MVO_MAXM], zs[MVO_MAXM], zsp[MVO_MAXM];
  for (int k = 0; k < M; ++k) {
    for (int j = 0; j < 3; ++j) zvec[k][j] = init_x[((size_t)tl.win * MVO_MAXM + k) * 4 + j];
    zt[k] = init_x[((size_t)tl.win * MVO_MAXM + k) * 4 + 3];
    zs[k] = d_scale(zt[k], prm.max_len, prm.min_len);
    zsp[k] = d_dscale(zt[k], prm.max_len, prm.min_len);
  }
  double u[3], nb;
  pair_unit_baseline(fr, M, (int)pd.a, (int)pd.b, zvec, zs, u, nb);
  const int a = pd.a, kmax = (int)pd.b < M - 1 ? (int)pd.b : M - 1;
  const double* __restrict__ R0 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f0;
  const double* __restrict__ R1 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f1;
  const int ncol = 3 * M;
  for (uint32_t i = lane; i < tl.count; i += 64) {
    const size_t o = (size_t)tl.src_start + i;
    const double* p0 = m0 + 3 * o;
    const double* p1 = m1 + 3 * o;
    const double px = R0[0] * p0[0] + R0[1] * p0[1] + R0[2] * p0[2];
    const double py = R0[3] * p0[0] + R0[4] * p0[1] + R0[5] * p0[2];
    const double pz = R0[6] * p0[0] + R0[7] * p0[1] + R0[8] * p0[2];
    const double qx = R1[0] * p1[0] + R1[1] * p1[1] + R1[2] * p1[2];
    const double qy = R1[3] * p1[0] + R1[4] * p1[1] + R1[5] * p1[2];
    const double qz = R1[6] * p1[0] + R1[7] * p1[1] + R1[8] * p1[2];
    const double cx = py * qz - pz * qy, cy = pz * qx - px * qz, cz = px * qy - py * qx;
    const double r = u[0] * cx + u[1] * cy + u[2] * cz;
    if (res) res[o] = r;
    if (jac) {
      double* __restrict__ row = jac + o * ncol;
      for (int c = 0; c < ncol; ++c) row[c] = 0.0;
      const double gx = (cx - u[0] * r) / nb, gy = (cy - u[1] * r) / nb, gz = (cz - u[2] * r) / nb;
      for (int k = a; k <= kmax; ++k) {
        const double vx = zvec[k][0], vy = zvec[k][1], vz = zvec[k][2];
        row[3 * k + 0] = zs[k] * (gx * (-2.0 * vy) + gy * (2.0 * vx));
        row[3 * k + 1] = zs[k] * (gx * (2.0 * vz) + gz * (-2.0 * vx));
        row[3 * k + 2] = zsp[k] * (vx * gx + vy * gy + vz * gz);
      }
    }
  }
}

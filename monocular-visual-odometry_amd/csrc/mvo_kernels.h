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
#include "mvo_wave.h"

#define FOLD_THREADS 256
#define FOLD_WAVES 4
#ifndef FOLD_BATCH
#define FOLD_BATCH 4
#endif
#ifndef PASS_BATCH
#define PASS_BATCH 1
#endif

// ---------------------------------------------------------------------------------------
// Work decomposition of the observation passes.  The host cuts the flat residual-block range
// into one CHUNK per wavefront (equal sizes, all wavefronts resident at once, so there is no
// tail round); a chunk is a list of TILES, each the intersection of the chunk with one frame
// pair.  wave_tiles[g] .. wave_tiles[g+1] are the tiles of wavefront g.  Per tile the wave
// reduces its sums once; the tiles of a workgroup that continue the same pair are merged through
// LDS into one SLOT (a slot never spans workgroups) -- no global atomics anywhere.
// ---------------------------------------------------------------------------------------
// `tiles` is the global tile table, or (persistent robust solver) the workgroup's own tiles copied to LDS: tiles[0] is
// then tile bt0, which `bias` = bt0 says.
__device__ __forceinline__ void block_store_partials(const TileDev* tiles, uint32_t bt0, uint32_t bt1,
                                                     const double* __restrict__ red,
                                                     double* __restrict__ partial, bool coherent, uint32_t bias = 0) {
  __syncthreads();
  tiles -= bias;
  const uint32_t nt = bt1 - bt0;
  for (uint32_t item = threadIdx.x; item < nt * 8; item += FOLD_THREADS) {
    const uint32_t lt = item >> 3, v = item & 7;
    const uint32_t slot = tiles[bt0 + lt].slot;
    if (lt == 0 || tiles[bt0 + lt - 1].slot != slot) {  // first tile of its run (runs are <= 4 long)
      double acc = red[lt * 8 + v];
      for (uint32_t q = lt + 1; q < nt && tiles[bt0 + q].slot == slot; ++q) acc += red[q * 8 + v];
      // coherent (a tail follows): device-coherent store, written through this XCD's L2 -- the workgroup
      // that adds the slots up (block_finish_pairs) may run on another XCD, and a release fence would
      // write back every dirty line of the L2, the c planes among them, once per workgroup.  Otherwise
      // a plain store: the next kernel finds the slots in the L2.
      if (coherent) __hip_atomic_store(partial + (size_t)slot * 8 + v, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else partial[(size_t)slot * 8 + v] = acc;
    }
  }
  // A tail follows: EVERY wavefront waits here until its own slot stores have been acknowledged by the
  // memory system (write-through stores are acknowledged once they are visible device-wide).  A
  // workgroup-scope release fence does not do that on gfx950 -- it emits no s_waitcnt vmcnt(0) -- and the
  // wavefront that takes the tickets (block_finish_pairs) only waits for its OWN stores; without this wait
  // another workgroup could draw a pair's last ticket and read a slot that wavefronts 1..3 of this
  // workgroup still have in flight.
  if (coherent) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One residual block of the evaluation pass.  ROBUST = ceres::CauchyLoss(0.5): b = 0.25, c = 4,
// rho(s) = b log(1 + c s), rho'(s) = 1 / (1 + c s); rho'' < 0 always, so Ceres' Corrector scales
// residual and Jacobian row by sqrt(rho') -- i.e. the moments are weighted by w = rho'.
// sum_i log(1 + c s_i) is carried as log(prod_i (1 + c s_i)): one multiply per block instead of
// one fp64 log (the factors lie in [1, 5]; the caller renormalises the product's exponent), and
// 1/(1 + c s) is a v_rcp_f64 seed plus two Newton steps (<= 1 ulp for arguments in [1, 5]).
template <bool ROBUST>
__device__ __forceinline__ void pass_accum(double cx, double cy, double cz, double u0, double u1, double u2,
                                           double (&S)[8], double& prod) {
  const double r = u0 * cx + u1 * cy + u2 * cz;
  const double s = r * r;
  double w = 1.0;
  if (ROBUST) {
    const double sum = fma(s, 4.0, 1.0);
    w = fmax(DBL_MIN, fast_rcp(sum));
    prod *= sum;
  } else {
    S[6] += s;
  }
  S[7] = fma(w, s, S[7]);
  const double wx = w * cx, wy = w * cy, wz = w * cz;
  S[0] = fma(wx, cx, S[0]);
  S[1] = fma(wx, cy, S[1]);
  S[2] = fma(wx, cz, S[2]);
  S[3] = fma(wy, cy, S[3]);
  S[4] = fma(wy, cz, S[4]);
  S[5] = fma(wz, cz, S[5]);
}

// ---------------------------------------------------------------------------------------
// Pass mode (a loss, or assembly_mode = PASSES): what an observation pass leaves for the LM step.
// k_lm publishes the unit baseline (u, |baseline|) of every pair at the point under evaluation; the
// workgroup that completes a pair (the ticket tails below) also PROJECTS the pair's sums.  A relaunched
// k_lm -- one evaluation per launch -- then reads ten numbers per pair and adds its classes up instead of
// running its per-pair stage (sqrt, reciprocal, ~100 flops per pair, up to eight pairs per lane) as ONE
// workgroup per window while the chip idles.  The projections are linear in the sums: in the
// observation-partitioned solve every rank projects its share, adds its classes up (k_class_sums) and the
// ranks all-reduce MVO_CLS_DOUBLES numbers per window (the "already-contracted" form).
// Measured and dropped (profiles/r3/NOTES.md): a second ticket level in the tails that also adds the
// classes up (three more dependent L2 round trips at the end of every pass), and passes that form the
// unit baselines themselves from a 20-number point (registers and a square root per tile).
// ---------------------------------------------------------------------------------------
constexpr int MVO_MCX = MVO_MAXM + 1;  // classes per frame a: j = min(b, M) in 0..M   (== MCX of mvo_lm.h)

struct PassOut {
  const double* upairs;  // [W * P][4]: (u, |baseline|) per pair at the point under evaluation
  double* proj;          // [W * P][10]: T(6) q(3) cost(1) per pair (project_pair8); nullptr: sums only
  int coherent_proj;     // the projections are read by other workgroups of THIS launch (persistent robust solver,
                         // k_lm<256, 6>): stored device-coherently, like the slots
  int coherent_sums;     // ... the pairs' sums are (fold and LM loop in one launch, k_lm<256, 7>)
};

// Projected moments of one frame pair from its 8 pass sums, by the 8 consecutive lanes that hold them
// (lane v of the group: sum v).  With u = the pair's unit baseline, nb its length and P = (I - u u^T) / nb:
//   o[0..5] = P S P (xx,xy,xz,yy,yz,zz)   o[6..8] = P S u   o[9] = 1/2 sum rho
// -- what k_lm's per-pair stage (pair_eval, mvo_lm.h) computes from the sums.  Lane v returns o[v]; lanes
// 0 and 1 also o[8], o[9] in `extra`.  Linear in S: ranks that hold a share of the pair's residual blocks
// project their share.
__device__ __forceinline__ double project_pair8(double acc, int lane, const double (&u)[3], double nb, double& extra) {
#pragma clang fp contract(off)
  const int base = lane & ~7, v = lane & 7;
  double S[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) S[j] = __shfl(acc, base + j, 64);
  const double u0 = u[0], u1 = u[1], u2 = u[2];
  const double inb = fast_rcp(nb), inb2 = inb * inb;
  const double Su0 = fma(S[2], u2, fma(S[1], u1, S[0] * u0));
  const double Su1 = fma(S[4], u2, fma(S[3], u1, S[1] * u0));
  const double Su2 = fma(S[5], u2, fma(S[4], u1, S[2] * u0));
  const double uSu = fma(u2, Su2, fma(u1, Su1, u0 * Su0));
  double o[10];
  o[0] = fma(u0 * u0, uSu, fma(-2.0 * u0, Su0, S[0])) * inb2;
  o[1] = fma(u0 * u1, uSu, fma(-Su0, u1, fma(-u0, Su1, S[1]))) * inb2;
  o[2] = fma(u0 * u2, uSu, fma(-Su0, u2, fma(-u0, Su2, S[2]))) * inb2;
  o[3] = fma(u1 * u1, uSu, fma(-2.0 * u1, Su1, S[3])) * inb2;
  o[4] = fma(u1 * u2, uSu, fma(-Su1, u2, fma(-u1, Su2, S[4]))) * inb2;
  o[5] = fma(u2 * u2, uSu, fma(-2.0 * u2, Su2, S[5])) * inb2;
  o[6] = fma(-u0, uSu, Su0) * inb;
  o[7] = fma(-u1, uSu, Su1) * inb;
  o[8] = fma(-u2, uSu, Su2) * inb;
  o[9] = 0.5 * S[6];
  double mine = o[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) mine = v == j ? o[j] : mine;
  extra = v ? o[9] : o[8];
  return mine;
}

// The 8 lanes holding pair p's sums (acc; lane v: sum v) project the pair and store the projection.
__device__ __forceinline__ void pair_project_store(const PassOut& po, uint32_t p, double acc, int lane) {
  const double* __restrict__ up = po.upairs + 4 * (size_t)p;
  const double u[3] = {up[0], up[1], up[2]};
  double extra;
  const double mine = project_pair8(acc, lane, u, up[3], extra);
  const int v = lane & 7;
  double* __restrict__ out = po.proj + 10 * (size_t)p;
  if (po.coherent_proj) {
    __hip_atomic_store(out + v, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < 2) __hip_atomic_store(out + 8 + v, extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    out[v] = mine;
    if (v < 2) out[8 + v] = extra;
  }
}

// The workgroup that stores the LAST slot of a frame pair adds that pair's slots up (what k_reduce_slots
// does as a launch of its own): k_lm then reads 8 sums per pair instead of walking slot lists, a chain
// of dependent L2 round trips per slot.  "Last" is decided by one ticket counter per pair -- the only
// global atomics of the pass, on counters, not on data: the slots are added in slot order whoever does
// it, bit for bit what k_reduce_slots gives.  No workgroup waits for another one.
// pair_ticket == nullptr: no tail (launches over a range of the workgroups, partitioned mode).
// `red` (the workgroup's reduction buffer, free after block_store_partials) holds the to-do list.
// po.proj != nullptr (the sums are those of a pass at the unit baselines po.upairs): the finishing workgroup
// also projects the pair (project_pair8).
// Returns the number of pairs this workgroup finished (the same value in every thread).
__device__ __forceinline__ uint32_t block_finish_pairs(int32_t* pair_ticket, const TileDev* tiles, uint32_t bt0,
                                                       uint32_t bt1, const PairDev* __restrict__ pairs, double* red,
                                                       const double* partial, double* __restrict__ pair_sums,
                                                       const PassOut& po, uint32_t bias = 0) {
  if (pair_ticket == nullptr) return 0;
  tiles -= bias;
  __shared__ uint32_t n_todo;
  // The slot stores above are device-coherent (written through the XCD's L2) and every wavefront has
  // waited for its own (s_waitcnt vmcnt(0) at the end of block_store_partials); the barrier below
  // therefore orders ALL slot stores of this workgroup before any ticket it takes.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (threadIdx.x == 0) n_todo = 0;
  __syncthreads();
  uint32_t* todo = reinterpret_cast<uint32_t*>(red);
  const uint32_t nt = bt1 - bt0;
  for (uint32_t lt = threadIdx.x; lt < nt; lt += FOLD_THREADS) {
    const uint32_t slot = tiles[bt0 + lt].slot;
    if (lt == 0 || tiles[bt0 + lt - 1].slot != slot) {  // one ticket per slot this workgroup stored
      const uint32_t p = tiles[bt0 + lt].pair;
      const PairDev pd = pairs[p];
      const int32_t t = __hip_atomic_fetch_add(pair_ticket + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (int32_t)(pd.slot_end - pd.slot_begin) - 1) {
        __hip_atomic_store(pair_ticket + p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
        todo[atomicAdd(&n_todo, 1u)] = p;
      }
    }
  }
  __syncthreads();
  const uint32_t nq = n_todo;
  const bool project = po.proj != nullptr;
  for (uint32_t item = threadIdx.x; item < nq * 8; item += FOLD_THREADS) {
    const uint32_t p = todo[item >> 3], v = item & 7;
    const PairDev pd = pairs[p];
    double acc = 0.0;
    for (uint32_t sl = pd.slot_begin; sl < pd.slot_end; ++sl)
      acc += __hip_atomic_load(partial + (size_t)sl * 8 + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (po.coherent_sums) __hip_atomic_store(pair_sums + (size_t)p * 8 + v, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else pair_sums[(size_t)p * 8 + v] = acc;
    // (the 8 items of a pair are 8 consecutive lanes, all inside the loop together: 8 | FOLD_THREADS)
    if (project) pair_project_store(po, p, acc, (int)(threadIdx.x & 63));
  }
  return nq;
}

// the tiles of wavefront gw: c = (R0 m0) x (R1 m1) per residual block, optionally stored, and the tile's 8 sums into
// red[(tile - bt0) * 8 + value].  Shared by k_fold and the fused fold + LM launch (k_lm<256, 7>).
template <bool WRITE_C, bool MOMENTS, int PASS0>
__device__ __forceinline__ void fold_wave_tiles(const TileDev* __restrict__ tiles, const uint32_t* __restrict__ wave_tiles,
                                                uint32_t gw, uint32_t n_waves, uint32_t bt0,
                                                const double* __restrict__ frames, const double* __restrict__ m0,
                                                const double* __restrict__ m1, double* __restrict__ cbuf, size_t cstride,
                                                double* red, int lane, const PassOut& po) {
  if (gw < n_waves) {
    const uint32_t t0 = wave_tiles[gw], t1 = wave_tiles[gw + 1];
    for (uint32_t t = t0; t < t1; ++t) {
      const TileDev tl = tiles[t];
      const double* __restrict__ R0 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f0;
      const double* __restrict__ R1 = frames + (size_t)MVO_FRAME_DOUBLES * tl.f1;
      const double a0 = R0[0], a1 = R0[1], a2 = R0[2], a3 = R0[3], a4 = R0[4], a5 = R0[5], a6 = R0[6],
                   a7 = R0[7], a8 = R0[8];
      const double b0 = R1[0], b1 = R1[1], b2 = R1[2], b3 = R1[3], b4 = R1[4], b5 = R1[5], b6 = R1[6],
                   b7 = R1[7], b8 = R1[8];
      double u0 = 0.0, u1 = 0.0, u2 = 0.0;
      if (PASS0) {
        u0 = po.upairs[4 * (size_t)tl.pair];
        u1 = po.upairs[4 * (size_t)tl.pair + 1];
        u2 = po.upairs[4 * (size_t)tl.pair + 2];
      }
      // (two rays per lane through 16-byte loads were tried: 7 % slower than one 8-byte load per component,
      // profiles/r2b/NOTES.md -- the fold runs at 5.1 TB/s, 81 % of what a plain copy reaches on this chip)
      const double* __restrict__ p0 = m0 + 3 * (size_t)tl.src_start;
      const double* __restrict__ p1 = m1 + 3 * (size_t)tl.src_start;
      double* __restrict__ cx = cbuf + tl.c_start;
      double* __restrict__ cy = cx + cstride;
      double* __restrict__ cz = cy + cstride;
      double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      double prod = 1.0;     // PASS0 == 2: running product of (1 + 4 r^2) of this lane
      int32_t prod_exp = 0;  // exponent split off so far
      // BATCH residual blocks per lane and trip, every load of a trip issued before the first use: a tile costs
      // ceil(count / (64 BATCH)) memory round trips (a remainder loop of single iterations costs one EACH).  Slots
      // past the tile's end re-read the lane's first block of the trip (an L1 hit) and are left out of the sums.
      // (64-bit indices: with 32-bit ones 3 * i may wrap as far as the compiler knows, and it then neither merges a
      // ray's three 8-byte loads into a 16- and an 8-byte one nor addresses the slots off one base)
      // (4, 5, 6, 7 slots measured alike at every size; more than four would push the variants that are also a pass
      // over 128 registers, and the one-round launch geometry assumes four wavefronts per SIMD)
      constexpr int BATCH = FOLD_BATCH;
      for (size_t i0 = lane; i0 < tl.count; i0 += 64 * 128) {
        const size_t i1 = (tl.count - i0) > 64 * 128 ? i0 + 64 * 128 : tl.count;
        for (size_t i = i0; i < i1; i += 64 * BATCH) {
          double x0[BATCH], y0[BATCH], z0[BATCH], x1[BATCH], y1[BATCH], z1[BATCH];
#pragma unroll
          for (int u = 0; u < BATCH; ++u) {
            const size_t j = i + 64 * u < i1 ? i + 64 * u : i;
            x0[u] = p0[3 * j]; y0[u] = p0[3 * j + 1]; z0[u] = p0[3 * j + 2];
            x1[u] = p1[3 * j]; y1[u] = p1[3 * j + 1]; z1[u] = p1[3 * j + 2];
          }
#pragma unroll
          for (int u = 0; u < BATCH; ++u) {
            const bool on = i + 64 * u < i1;
            const double px = a0 * x0[u] + a1 * y0[u] + a2 * z0[u];  // R0 * m0
            const double py = a3 * x0[u] + a4 * y0[u] + a5 * z0[u];
            const double pz = a6 * x0[u] + a7 * y0[u] + a8 * z0[u];
            const double qx = b0 * x1[u] + b1 * y1[u] + b2 * z1[u];  // R1 * m1
            const double qy = b3 * x1[u] + b4 * y1[u] + b5 * z1[u];
            const double qz = b6 * x1[u] + b7 * y1[u] + b8 * z1[u];
            const double vx = py * qz - pz * qy;  // c = p x q
            const double vy = pz * qx - px * qz;
            const double vz = px * qy - py * qx;
            if (on) {
              if (WRITE_C) {
                cx[i + 64 * u] = vx;
                cy[i + 64 * u] = vy;
                cz[i + 64 * u] = vz;
              }
              if (MOMENTS) {
                S[0] += vx * vx;
                S[1] += vx * vy;
                S[2] += vx * vz;
                S[3] += vy * vy;
                S[4] += vy * vz;
                S[5] += vz * vz;
              }
              if (PASS0) pass_accum<PASS0 == 2>(vx, vy, vz, u0, u1, u2, S, prod);
            }
          }
        }
        if (PASS0 == 2) {  // <= 128 factors <= 5 since the last split: far from overflow
          int e;
          prod = frexp(prod, &e);
          prod_exp += e;
        }
      }
      if (PASS0 == 2) S[6] = 0.25 * (log(prod) + (double)prod_exp * 0.693147180559945309417232121458);
      if (MOMENTS || PASS0) {
        const double tot = wave_reduce8(S, lane);
        if ((lane & 7) == 0) red[(t - bt0) * 8 + reduce8_value_of_lane(lane)] = tot;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_fold: reads m0, m1 (48 B / residual block, caller's interleaved layout), optionally writes
// c (24 B / block) and/or accumulates S = sum c c^T.
// ---------------------------------------------------------------------------------------
// PASS0 (robust path): besides storing c, the fold is also the evaluation pass at the starting point --
// u(x0) per pair is known before it starts (pair_u0, computed by the host at upload), c is in registers:
// one of the max_it + 1 passes over c saved.  1: w = 1, 2: Cauchy(0.5); sums as in k_pass.
template <bool WRITE_C, bool MOMENTS, int PASS0 = 0>
__global__ __launch_bounds__(FOLD_THREADS) void k_fold(const TileDev* __restrict__ tiles,
                                                       const uint32_t* __restrict__ wave_tiles, uint32_t n_waves,
                                                       const double* __restrict__ frames,
                                                       const double* __restrict__ m0,
                                                       const double* __restrict__ m1,
                                                       double* __restrict__ cbuf, size_t cstride,
                                                       double* partial, int32_t* pair_ticket,
                                                       const PairDev* __restrict__ pairs,
                                                       double* __restrict__ pair_sums, PassOut po) {
  static_assert(!(MOMENTS && PASS0), "either the plain moments or the pass sums");
  extern __shared__ double red[];  // [tiles of this workgroup][8]
  const int lane = threadIdx.x & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t bw0 = blockIdx.x * FOLD_WAVES;
  const uint32_t bw1 = bw0 + FOLD_WAVES < n_waves ? bw0 + FOLD_WAVES : n_waves;
  const uint32_t bt0 = wave_tiles[bw0], bt1 = wave_tiles[bw1];
  const uint32_t gw = bw0 + wv;
  fold_wave_tiles<WRITE_C, MOMENTS, PASS0>(tiles, wave_tiles, gw, n_waves, bt0, frames, m0, m1, cbuf, cstride, red, lane, po);
  if (MOMENTS || PASS0) {
    block_store_partials(tiles, bt0, bt1, red, partial, pair_ticket != nullptr);
    block_finish_pairs(pair_ticket, tiles, bt0, bt1, pairs, red, partial, pair_sums, po);
  }
}

// ---------------------------------------------------------------------------------------
// k_pass: one evaluation of all residual blocks at the unit baselines in pair_u.
// reads c (24 B / block) as 16-byte loads from three SoA planes.
// partial[slot] = { S (6, weighted), sum rho(r^2) or sum r^2, sum w r^2 }
// ---------------------------------------------------------------------------------------
// the tiles of wavefront gw: one evaluation of their residual blocks at the unit baselines in `upairs`; the tile's
// 8 sums land in red[(tile - bt0) * 8 + value].  Shared by k_pass and the persistent robust solver (k_lm<256, 6>).
// (tiles / bias: see block_store_partials; wave_tiles[0] is then the entry of wavefront `wbias`)
template <bool ROBUST>
__device__ __forceinline__ void pass_wave_tiles(const TileDev* tiles, const uint32_t* wave_tiles,
                                                uint32_t gw, uint32_t n_waves, uint32_t bt0,
                                                const double* __restrict__ cbuf, size_t cstride,
                                                const int32_t* __restrict__ win_status,  // nullptr: every window is live
                                                const double* upairs, double* red, int lane, uint32_t bias = 0,
                                                uint32_t wbias = 0) {
  if (gw >= n_waves) return;
  tiles -= bias;
  const uint32_t t0 = wave_tiles[gw - wbias], t1 = wave_tiles[gw + 1 - wbias];
  for (uint32_t t = t0; t < t1; ++t) {
    const TileDev tl = tiles[t];
    double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (win_status == nullptr || win_status[tl.win] == 0) {
      const double u0 = upairs[4 * (size_t)tl.pair], u1 = upairs[4 * (size_t)tl.pair + 1],
                   u2 = upairs[4 * (size_t)tl.pair + 2];
      // 16-byte loads need an even start: step back one element when the tile starts odd
      const uint32_t head = tl.c_start & 1u;
      const double* __restrict__ cx = cbuf + (tl.c_start - head);
      const double* __restrict__ cy = cx + cstride;
      const double* __restrict__ cz = cy + cstride;
      const uint32_t len = tl.count + head;
      double prod = 1.0;     // running product of (1 + 4 r^2) of this lane
      int32_t prod_exp = 0;  // exponent split off so far
      // PASS_BATCH x two elements per lane and trip, every load of a trip issued before the first use (see k_fold);
      // slots past the tile's end re-read the lane's first pair of the trip and enter as zeros (no-ops of pass_accum)
      for (size_t i0 = 2 * lane; i0 < len; i0 += 128 * 64) {
        const size_t i1 = (len - i0) > 128 * 64 ? i0 + 128 * 64 : len;
        for (size_t i = i0; i < i1; i += 128 * PASS_BATCH) {
          double2 x[PASS_BATCH], y[PASS_BATCH], z[PASS_BATCH];
#pragma unroll
          for (int u = 0; u < PASS_BATCH; ++u) {
            const size_t j = i + 128 * u < i1 ? i + 128 * u : i;
            x[u] = *reinterpret_cast<const double2*>(cx + j);
            y[u] = *reinterpret_cast<const double2*>(cy + j);
            z[u] = *reinterpret_cast<const double2*>(cz + j);
          }
#pragma unroll
          for (int u = 0; u < PASS_BATCH; ++u) {
            const size_t j = i + 128 * u;
            const bool v0 = j >= head && j < i1, v1 = (j + 1) < len;  // neighbours' elements and pads are not data
            pass_accum<ROBUST>(v0 ? x[u].x : 0.0, v0 ? y[u].x : 0.0, v0 ? z[u].x : 0.0, u0, u1, u2, S, prod);
            pass_accum<ROBUST>(v1 ? x[u].y : 0.0, v1 ? y[u].y : 0.0, v1 ? z[u].y : 0.0, u0, u1, u2, S, prod);
          }
        }
        if (ROBUST) {  // <= 128 factors <= 5 since the last split: far from overflow
          int e;
          prod = frexp(prod, &e);
          prod_exp += e;
        }
      }
      if (ROBUST) S[6] = 0.25 * (log(prod) + (double)prod_exp * 0.693147180559945309417232121458);
    }
    const double tot = wave_reduce8(S, lane);
    if ((lane & 7) == 0) red[(t - bt0) * 8 + reduce8_value_of_lane(lane)] = tot;
  }
}

template <bool ROBUST>
__global__ __launch_bounds__(FOLD_THREADS) void k_pass(const TileDev* __restrict__ tiles,
                                                       const uint32_t* __restrict__ wave_tiles, uint32_t n_waves,
                                                       const double* __restrict__ cbuf, size_t cstride,
                                                       const int32_t* __restrict__ win_status,
                                                       double* partial, int32_t* pair_ticket,
                                                       const PairDev* __restrict__ pairs,
                                                       double* __restrict__ pair_sums, PassOut po) {
  extern __shared__ double red[];
  const int lane = threadIdx.x & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t bw0 = blockIdx.x * FOLD_WAVES;
  const uint32_t bw1 = bw0 + FOLD_WAVES < n_waves ? bw0 + FOLD_WAVES : n_waves;
  const uint32_t bt0 = wave_tiles[bw0], bt1 = wave_tiles[bw1];
  pass_wave_tiles<ROBUST>(tiles, wave_tiles, bw0 + wv, n_waves, bt0, cbuf, cstride, win_status, po.upairs, red, lane);
  block_store_partials(tiles, bt0, bt1, red, partial, pair_ticket != nullptr);
  block_finish_pairs(pair_ticket, tiles, bt0, bt1, pairs, red, partial, pair_sums, po);
}

// ---------------------------------------------------------------------------------------
// k_reduce_slots: per-pair sums of the slot partials (observation-partitioned mode: these 8
// numbers per pair are what the ranks all-reduce before the identical LM step).
// ---------------------------------------------------------------------------------------
// po.proj != nullptr: also the pairs' projected moments -- the launch-of-its-own form of the passes' tails
// (MVO_HIP_NO_TAILS: the tails' bit-for-bit reference).
__global__ __launch_bounds__(256) void k_reduce_slots(const PairDev* __restrict__ pairs, uint32_t n_pairs,
                                                      const double* __restrict__ partial,
                                                      double* __restrict__ pair_sums, PassOut po) {
  const uint32_t item = blockIdx.x * 256 + threadIdx.x;
  if (item >= n_pairs * 8) return;  // (whole groups of 8 lanes leave together)
  const uint32_t p = item >> 3, v = item & 7;
  const PairDev pd = pairs[p];
  double acc = 0.0;
  for (uint32_t sl = pd.slot_begin; sl < pd.slot_end; ++sl) acc += partial[(size_t)sl * 8 + v];
  pair_sums[item] = acc;
  if (po.proj != nullptr) pair_project_store(po, p, acc, (int)(threadIdx.x & 63));
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

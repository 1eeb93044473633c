"""Multi-GPU partition plans for the refinement path.

* independent windows (BASELINE config 3): a static block partition of window ids, one process
  per GPU, no data-path collective;
* one large window (config 4): every pair's residual blocks are range-partitioned; the per-pair
  sums are additive, so one all-reduce of 8 doubles per pair precedes each (identical) LM step.
"""
from __future__ import annotations

import numpy as np

from . import scene


def shard_range(n: int, rank: int, world: int):
    """[lo, hi) of a balanced contiguous partition of range(n)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_windows(n_windows: int, rank: int, world: int):
    lo, hi = shard_range(n_windows, rank, world)
    return range(lo, hi)


def shard_observations(w: "scene.Window", rank: int, world: int) -> "scene.Window":
    """This rank's share of every frame pair's residual blocks (same frames, same pair order)."""
    offs = [0]
    a0, a1 = [], []
    for p in range(w.n_pairs):
        lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
        s0, s1 = shard_range(hi - lo, rank, world)
        a0.append(w.m0[lo + s0:lo + s1])
        a1.append(w.m1[lo + s0:lo + s1])
        offs.append(offs[-1] + (s1 - s0))
    return scene.Window(
        N=w.N, M=w.M, frames=w.frames, pair_offsets=np.array(offs, dtype=np.uint64),
        m0=np.ascontiguousarray(np.concatenate(a0)) if a0 else w.m0[:0],
        m1=np.ascontiguousarray(np.concatenate(a1)) if a1 else w.m1[:0],
        true_vec=w.true_vec, true_scale=w.true_scale, bearing_observations=w.bearing_observations,
        meta=dict(w.meta, shard=(rank, world)),
    )


CLASS_DOUBLES = 352  # include/mvo_hip.h: MVO_HIP_CLASS_DOUBLES


def contracted_class_totals(w: "scene.Window", use_loss: bool) -> np.ndarray:
    """Numpy statement (host logic of the plan, used by the tests) of what the ranks all-reduce per
    evaluation in the partitioned solve WITH a loss, at the window's starting point: per frame pair the
    projected moments T = P S P (xx,xy,xz,yy,yz,zz), q = P S u and 1/2 sum rho, with S = sum_i w_i c_i c_i^T,
    u the pair's unit baseline, P = (I - u u^T) / |baseline|, summed into classes
    (a, j = min(b, M)); layout cls[5][6][10] then kk[5][10] (the pair (k, k) alone), 352 doubles.  Linear in
    S, hence additive over any partition of the residual blocks."""
    M, N = w.M, w.N
    R = w.frames[:, :9].reshape(N, 3, 3)
    vec, scale = w.frames[:, 9:12], w.frames[:, 12]
    out = np.zeros(CLASS_DOUBLES)
    cls, kk = out[:300].reshape(5, 6, 10), out[300:350].reshape(5, 10)
    for p, (a, b) in enumerate(scene.pair_list(M, N)):
        lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
        c = np.cross(w.m0[lo:hi] @ R[a].T, w.m1[lo:hi] @ R[b + 1].T)
        beta = sum(scale[k] * vec[k] for k in range(a, min(b, M - 1) + 1))
        if b >= M:  # IterativeRefinement.cpp:137-144: only frame b itself
            beta = beta + scale[b] * vec[b]
        nb = np.linalg.norm(beta)
        u = beta / nb
        r = c @ u
        if use_loss:  # ceres::CauchyLoss(0.5)
            wgt, rho = 1.0 / (1.0 + 4.0 * r * r), 0.25 * np.log1p(4.0 * r * r)
        else:
            wgt, rho = np.ones_like(r), r * r
        S = (c * wgt[:, None]).T @ c
        P = (np.eye(3) - np.outer(u, u)) / nb
        T, q = P @ S @ P, P @ S @ u
        o = np.array([T[0, 0], T[0, 1], T[0, 2], T[1, 1], T[1, 2], T[2, 2], q[0], q[1], q[2], 0.5 * rho.sum()])
        cls[a, min(b, M)] += o
        if a == b:
            kk[a] = o
    return out


def shard_landmarks(sc, rank: int, world: int):
    """Bundle-adjustment scene (ba_scene.BaScene) restricted to this rank's contiguous range of
    landmarks, cut so that every rank gets about the same number of observations; all poses stay.
    Returns (shard, (lo, hi)) with lo/hi the landmark range in the full scene."""
    from dataclasses import replace

    offs = np.asarray(sc.lm_offsets, dtype=np.uint64)
    n_obs = int(offs[-1])
    cuts = [int(np.searchsorted(offs, (n_obs * r) // world, side="left")) for r in range(world + 1)]
    L = len(offs) - 1
    if world > L:
        # mvo_hip_ba_upload refuses an empty landmark range; raised on EVERY rank alike (the plan
        # depends on the scene and `world` only), i.e. before any rank enters a collective
        raise ValueError(f"cannot partition {L} landmarks over {world} ranks: every rank needs at least one")
    cuts[0], cuts[-1] = 0, L
    for r in range(1, world):  # every rank owns at least one landmark, however skewed the track lengths
        cuts[r] = min(max(cuts[r], cuts[r - 1] + 1), L - (world - r))
    lo, hi = cuts[rank], cuts[rank + 1]
    o0, o1 = int(offs[lo]), int(offs[hi])
    shard = replace(sc, X=np.ascontiguousarray(sc.X[lo:hi]),
                    lm_offsets=np.ascontiguousarray(offs[lo:hi + 1] - offs[lo]),
                    obs_frame=np.ascontiguousarray(sc.obs_frame[o0:o1]),
                    obs_ray=np.ascontiguousarray(sc.obs_ray[o0:o1]), true_X=sc.true_X[lo:hi],
                    meta=dict(sc.meta, shard=(rank, world)))
    return shard, (lo, hi)

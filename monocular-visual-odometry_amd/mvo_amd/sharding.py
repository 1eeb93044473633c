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

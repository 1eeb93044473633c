"""Seeded synthetic scenes for the explicit-landmark window bundle adjustment.

Same trajectory / frustum recipe as scene.py: frame 0 is the newest keyframe, R_i is
camera-to-world (RefinementFrame.R's convention, bearing = R_i^T (X - p_i)), landmarks sit
4..40 m ahead in a 60 x 45 degree frustum, bearings carry tangent-plane Gaussian noise.  The
newest M frames and all landmarks are free; frames M..N-1 anchor the gauge.  No counterpart in the
reference (it has no landmarks): everything here is synthetic by construction.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .scene import _small_rotation


@dataclass
class BaScene:
    N: int
    M: int
    poses: np.ndarray        # [N, 12] float64: R(9) row-major camera-to-world, p(3)  -- initial guess
    X: np.ndarray            # [L, 3] initial landmark positions
    lm_offsets: np.ndarray   # [L+1] uint64: observations of landmark j are [off[j], off[j+1])
    obs_frame: np.ndarray    # [n_obs] uint32
    obs_ray: np.ndarray      # [n_obs, 3] unit bearing rays in the camera frame
    true_poses: np.ndarray
    true_X: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def n_landmarks(self) -> int:
        return len(self.X)

    @property
    def n_observations(self) -> int:
        return int(self.lm_offsets[-1])


def make_ba_scene(N: int, M: int, n_landmarks: int, seed: int, span: int | None = None,
                  noise_sigma: float = 1e-3, outlier_fraction: float = 0.0,
                  pose_rot_deg: float = 0.3, pose_pos_sigma: float = 0.03,
                  landmark_sigma_rel: float = 0.02) -> BaScene:
    """span = frames each landmark is seen in (contiguous; None = all N)."""
    assert 1 <= M < N
    rng = np.random.Generator(np.random.MT19937(seed))
    step = rng.uniform(0.4, 1.6, N)
    d = np.stack([0.35 + rng.uniform(0.0, 0.3, N), rng.normal(0.0, 0.1, N), np.ones(N)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = np.zeros((N, 3))
    for i in range(1, N):
        pos[i] = pos[i - 1] + step[i - 1] * d[i - 1]
    R = np.empty((N, 3, 3))
    R[0] = _small_rotation(rng, np.deg2rad(2.0))
    for i in range(1, N):
        R[i] = _small_rotation(rng, np.deg2rad(2.0)) @ R[i - 1]

    L = n_landmarks
    ln = N if span is None else int(min(span, N))
    lo = rng.integers(0, N - ln + 1, L)
    hi = lo + ln - 1
    z = rng.uniform(4.0, 40.0, L)
    cam = np.stack([z * np.tan(np.deg2rad(rng.uniform(-30.0, 30.0, L))),
                    z * np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, L))), z], axis=1)
    X = pos[hi] + np.einsum("lij,lj->li", R[hi], cam)

    counts = (hi - lo + 1).astype(np.uint64)
    offs = np.zeros(L + 1, dtype=np.uint64)
    np.cumsum(counts, out=offs[1:])
    n_obs = int(offs[-1])
    lm_of_obs = np.repeat(np.arange(L), counts.astype(np.int64))
    frame = (np.arange(n_obs) - np.repeat(offs[:-1].astype(np.int64), counts.astype(np.int64))
             + np.repeat(lo, counts.astype(np.int64))).astype(np.uint32)
    v = np.einsum("oji,oj->oi", R[frame], X[lm_of_obs] - pos[frame])  # R^T (X - p)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    if noise_sigma > 0:
        t1 = np.cross(v, np.array([0.0, 1.0, 0.0]))
        t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
        t2 = np.cross(v, t1)
        e = rng.normal(0.0, noise_sigma, (n_obs, 2))
        v = v + e[:, :1] * t1 + e[:, 1:] * t2
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    if outlier_fraction > 0:
        bad = rng.random(n_obs) < outlier_fraction
        nb = int(bad.sum())
        r = np.stack([np.tan(np.deg2rad(rng.uniform(-30.0, 30.0, nb))),
                      np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, nb))), np.ones(nb)], axis=1)
        v[bad] = r / np.linalg.norm(r, axis=1, keepdims=True)

    true_poses = np.concatenate([R.reshape(N, 9), pos], axis=1)
    poses = true_poses.copy()
    for k in range(M):
        poses[k, :9] = (R[k] @ _small_rotation(rng, np.deg2rad(pose_rot_deg))).reshape(9)
        poses[k, 9:] += rng.normal(0.0, pose_pos_sigma, 3)
    X0 = X + rng.normal(0.0, 1.0, (L, 3)) * (landmark_sigma_rel * z)[:, None]
    return BaScene(N=N, M=M, poses=np.ascontiguousarray(poses), X=np.ascontiguousarray(X0), lm_offsets=offs,
                   obs_frame=np.ascontiguousarray(frame), obs_ray=np.ascontiguousarray(v),
                   true_poses=true_poses, true_X=X,
                   meta=dict(seed=seed, span=span, noise_sigma=noise_sigma, outlier_fraction=outlier_fraction))


def ba_config1(seed: int = 2001) -> BaScene:
    """BASELINE.json configs[0] shape: 10 keyframes, 300 landmarks, all visible: 3 000 observations."""
    return make_ba_scene(10, 8, 300, seed)


def ba_config2(seed: int = 2002, outlier_fraction: float = 0.0) -> BaScene:
    """configs[1] shape: 30 keyframes, 5 000 landmarks seen in 20 frames each: 100 000 observations."""
    return make_ba_scene(30, 28, 5000, seed, span=20, outlier_fraction=outlier_fraction)


def ba_config4(seed: int = 2004, n_landmarks: int = 200_000, outlier_fraction: float = 0.0) -> BaScene:
    """configs[3] shape scaled to the reduced-camera-system limit of this round (28 free frames):
    200 keyframes of which the newest 28 are free, 200 000 landmarks x 50 frames = 10 M observations."""
    return make_ba_scene(200, 28, n_landmarks, seed, span=50, outlier_fraction=outlier_fraction)

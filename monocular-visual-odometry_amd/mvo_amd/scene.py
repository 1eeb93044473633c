"""Seeded synthetic scene / trajectory generator for the sliding-window refinement path.

Produces exactly what the reference's ``IterativeRefinement::refine`` consumes
(/root/reference/mvo/src/algorithms/IterativeRefinement.cpp:15-183):

* ``frames``  -- N records (R row-major 3x3, unit baseline ``vec`` to the previous frame,
  ``scale``), index 0 = newest (Refiner.cpp:62-68);
* residual blocks in the reference's pair order ``for a < M: for b in [a, N-1)``
  (IterativeRefinement.cpp:116,134), CSR ``pair_offsets`` plus unit rays ``m0`` (frame a) and
  ``m1`` (frame b+1) as ``Frame::getCorrespondingFeatures`` + ``normFeatures`` would return
  them (Frame.hpp:258-302, FeatureOperations.cpp:117-122): a track contributes to pair (a, b)
  iff it is alive in frame a and reaches back to frame b+1.

The recipe follows SURVEY.md section 8(d).  Everything is deterministic in ``seed``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

LOWEST_LENGTH = 0.25  # mvo/cfg/mvo.cfg:45
HIGHEST_LENGTH = 2.0  # mvo/cfg/mvo.cfg:46


def num_pairs(M: int, N: int) -> int:
    return M * (N - 1) - M * (M - 1) // 2


def pair_list(M: int, N: int):
    return [(a, b) for a in range(M) for b in range(a, N - 1)]


@dataclass
class Window:
    N: int
    M: int
    frames: np.ndarray  # [N, 13] float64: R(9) vec(3) scale(1)  (== the C-ABI frame record)
    pair_offsets: np.ndarray  # [P+1] uint64
    m0: np.ndarray  # [n_obs, 3] float64 unit rays in frame a
    m1: np.ndarray  # [n_obs, 3] float64 unit rays in frame b+1
    true_vec: np.ndarray  # [N, 3]
    true_scale: np.ndarray  # [N]
    bearing_observations: int = 0
    meta: dict = field(default_factory=dict)

    @property
    def n_pairs(self) -> int:
        return len(self.pair_offsets) - 1

    @property
    def n_residual_blocks(self) -> int:
        return int(self.pair_offsets[-1])


def _small_rotation(rng, sigma_rad):
    w = rng.normal(0.0, sigma_rad, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-300:
        return np.eye(3)
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def _rotate_about_random_axis(rng, v, angle_rad):
    """Rotate unit vector v by angle about a random axis perpendicular to it."""
    r = rng.normal(size=3)
    axis = np.cross(v, r)
    axis /= np.linalg.norm(axis)
    return v * np.cos(angle_rad) + np.cross(axis, v) * np.sin(angle_rad)


def make_window(
    N: int,
    M: int,
    n_landmarks: int,
    seed: int,
    span="full",
    noise_sigma: float = 1e-3,
    outlier_fraction: float = 0.0,
    init_angle_deg: float = 3.0,
    init_scale_jitter: float = 0.2,
    min_abs_vx: float = 0.2,
    keep_tracks: bool = False,
    direction=None,
) -> Window:
    """Build one window.

    span: "full" (every landmark seen in every frame) |
          ("anchored", lo_span, hi_span): alive from the newest frame, length U[lo,hi] |
          ("sliding", length): contiguous spans of `length` frames, start uniform so that about
          half of the landmarks are alive in the newest M frames.
    direction: None = the parity recipe (|v.x| >= 0.2); (axis, vx) = the chart-singularity stress
          recipe of SURVEY.md section 7: every frame moves along `axis` ("z": forward motion (0,0,1),
          "y": (0,1,0)) with the x component of the true AND of the initial direction set to exactly
          `vx` -- J_loc = [[-2y, 2z], [2x, 0], [0, -2x]] (IterativeRefinement.cpp:282-296,334-346) has
          rank 1 at x = 0.  The initial guess is the true direction turned about the x axis, which
          leaves x untouched.
    """
    assert 1 <= M < N
    rng = np.random.Generator(np.random.MT19937(seed))

    # --- trajectory -------------------------------------------------------------------
    true_scale = rng.uniform(0.4, 1.6, N)
    d = np.stack(
        [0.35 + rng.uniform(0.0, 0.3, N), rng.normal(0.0, 0.1, N), np.ones(N)], axis=1
    )
    true_vec = d / np.linalg.norm(d, axis=1, keepdims=True)
    if direction is not None:
        axis, vx = direction
        rest = np.sqrt(1.0 - vx * vx)
        true_vec = np.tile(np.array([vx, 0.0, rest] if axis == "z" else [vx, rest, 0.0]), (N, 1))
    else:
        assert np.all(np.abs(true_vec[:, 0]) >= min_abs_vx - 1e-12)
    pos = np.zeros((N + 1, 3))
    for i in range(N):
        pos[i + 1] = pos[i] + true_scale[i] * true_vec[i]
    R = np.empty((N, 3, 3))
    R[0] = _small_rotation(rng, np.deg2rad(2.0))
    for i in range(1, N):
        R[i] = _small_rotation(rng, np.deg2rad(2.0)) @ R[i - 1]

    # --- landmark visibility spans [lo, hi] (frame indices, lo = newest) -----------------
    L = n_landmarks
    if span == "full":
        lo = np.zeros(L, dtype=np.int64)
        hi = np.full(L, N - 1, dtype=np.int64)
    elif span[0] == "anchored":
        lo = np.zeros(L, dtype=np.int64)
        ln = rng.integers(span[1], span[2] + 1, L)
        hi = np.minimum(lo + ln - 1, N - 1)
    elif span[0] == "sliding":
        ln = int(span[1])
        lo = rng.integers(0, max(1, min(2 * M, N - ln) + 1), L)
        hi = np.minimum(lo + ln - 1, N - 1)
    else:
        raise ValueError(span)
    bearing_obs = int(np.sum(hi - lo + 1))
    active = np.nonzero((lo < M) & (hi > lo))[0]
    lo_a, hi_a = lo[active], hi[active]
    La = len(active)

    # --- landmarks: frustum 60 x 45 deg, depth U[4,40] ahead of the oldest frame of the span
    z = rng.uniform(4.0, 40.0, La)
    xc = z * np.tan(np.deg2rad(rng.uniform(-30.0, 30.0, La)))
    yc = z * np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, La)))
    cam = np.stack([xc, yc, z], axis=1)
    X = pos[hi_a] + np.einsum("lij,lj->li", R[hi_a], cam)

    # --- bearings with tangent-plane noise, one per (frame, landmark) --------------------
    rays = np.empty((N, La, 3))
    for i in range(N):
        v = (X - pos[i]) @ R[i]  # R_i^T (X - p_i)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        if noise_sigma > 0:
            t1 = np.cross(v, np.array([0.0, 1.0, 0.0]))
            t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
            t2 = np.cross(v, t1)
            e = rng.normal(0.0, noise_sigma, (La, 2))
            v = v + e[:, :1] * t1 + e[:, 1:] * t2
            v /= np.linalg.norm(v, axis=1, keepdims=True)
        rays[i] = v

    # --- residual blocks in reference pair order ---------------------------------------
    pairs = pair_list(M, N)
    offs = np.zeros(len(pairs) + 1, dtype=np.uint64)
    m0_parts, m1_parts = [], []
    for p, (a, b) in enumerate(pairs):
        ids = np.nonzero((lo_a <= a) & (hi_a >= b + 1))[0]
        m0_parts.append(rays[a, ids])
        m1_parts.append(rays[b + 1, ids])
        offs[p + 1] = offs[p] + np.uint64(len(ids))
    m0 = np.ascontiguousarray(np.concatenate(m0_parts, axis=0)) if m0_parts else np.zeros((0, 3))
    m1 = np.ascontiguousarray(np.concatenate(m1_parts, axis=0)) if m1_parts else np.zeros((0, 3))

    if outlier_fraction > 0 and len(m1):
        n = len(m1)
        bad = rng.random(n) < outlier_fraction
        nb = int(bad.sum())
        zz = np.ones(nb)
        xx = np.tan(np.deg2rad(rng.uniform(-30.0, 30.0, nb)))
        yy = np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, nb)))
        r = np.stack([xx, yy, zz], axis=1)
        m1[bad] = r / np.linalg.norm(r, axis=1, keepdims=True)

    # --- initial guess ----------------------------------------------------------------
    frames = np.zeros((N, 13))
    frames[:, :9] = R.reshape(N, 9)
    frames[:, 9:12] = true_vec
    frames[:, 12] = true_scale
    for k in range(M):
        frames[k, 9:12] = _rotate_about_random_axis(rng, true_vec[k], np.deg2rad(init_angle_deg))
        if direction is not None:  # turn about the x axis: the x component stays exactly vx
            ang = np.deg2rad(init_angle_deg) * (1.0 if rng.random() < 0.5 else -1.0)
            c, sn = np.cos(ang), np.sin(ang)
            v = true_vec[k]
            r = np.array([0.0, c * v[1] - sn * v[2], sn * v[1] + c * v[2]])
            r *= np.sqrt(1.0 - v[0] * v[0]) / np.linalg.norm(r)
            frames[k, 9:12] = [v[0], r[1], r[2]]
        s = true_scale[k] * rng.uniform(1.0 - init_scale_jitter, 1.0 + init_scale_jitter)
        frames[k, 12] = float(np.clip(s, LOWEST_LENGTH + 0.01, HIGHEST_LENGTH - 0.01))

    meta = dict(seed=seed, n_landmarks=n_landmarks, span=span, noise_sigma=noise_sigma,
                outlier_fraction=outlier_fraction)
    if keep_tracks:
        assert outlier_fraction == 0, "outliers are injected per residual block, not per track"
        meta["tracks"] = dict(lo=lo_a, hi=hi_a, rays=rays)
    return Window(
        N=N, M=M, frames=frames, pair_offsets=offs, m0=m0, m1=m1, true_vec=true_vec,
        true_scale=true_scale, bearing_observations=bearing_obs, meta=meta,
    )


def track_frames(w: Window, unnormalised: bool = True):
    """The window (built with keep_tracks=True) as the reference's per-frame Feature tables
    (sliding_window/Feature.hpp:6-22), OLDEST FRAME FIRST, as a tracker would hand them over:
    [(rays[n,3], pre_feature[n] int32)], features in landmark order, pre_feature = index of the same
    landmark in the next older frame or -1.  unnormalised: rays as projectPixelTo3dRay returns them,
    (x, y, 1) (FeatureOperations.cpp:31-33), otherwise the unit rays."""
    tr = w.meta["tracks"]
    lo, hi, rays = tr["lo"], tr["hi"], tr["rays"]
    N = w.N
    ids = [np.nonzero((lo <= i) & (hi >= i))[0] for i in range(N)]
    out = []
    for i in range(N - 1, -1, -1):
        r = rays[i, ids[i]]
        if unnormalised:
            r = r / r[:, 2:3]
        pre = -np.ones(len(ids[i]), dtype=np.int32)
        if i + 1 < N:
            index = -np.ones(len(lo), dtype=np.int64)
            index[ids[i + 1]] = np.arange(len(ids[i + 1]))
            pre = index[ids[i]].astype(np.int32)
        out.append((np.ascontiguousarray(r), pre))
    return out


def gather_tracks_numpy(frames_oldest_first, M: int):
    """Numpy restatement of the reference's host-side gather for test comparison (chain walk of
    Frame.hpp:277-291 per pair of IterativeRefinement.cpp:116-152, normFeatures
    FeatureOperations.cpp:117-122) -> (pair_offsets, m0, m1)."""
    fr = frames_oldest_first[::-1]  # index 0 = newest
    N = len(fr)
    counter = [None] * N
    for i in range(N - 1, -1, -1):  # Frame.cpp:132-148
        pre = fr[i][1]
        if i == N - 1:
            counter[i] = np.zeros(len(pre), dtype=np.int64)
        else:
            ok = (pre >= 0) & (pre < len(fr[i + 1][1]))
            older = np.concatenate([counter[i + 1], [0]])  # one pad entry: indexable when empty
            counter[i] = np.where(ok, older[np.where(ok, pre, 0)] + 1, 0)
    offs = [0]
    m0s, m1s = [], []
    for a, b in pair_list(M, N):
        past = b - a + 1
        sel = np.nonzero(counter[a] >= past)[0]
        idx = sel.copy()
        for s in range(past):
            idx = fr[a + s][1][idx]
        v0 = fr[a][0][sel]
        v1 = fr[b + 1][0][idx]
        for v, dst in ((v0, m0s), (v1, m1s)):
            scale = 1.0 / np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
            dst.append(v * scale[:, None])
        offs.append(offs[-1] + len(sel))
    cat = lambda parts: np.ascontiguousarray(np.concatenate(parts, axis=0)) if parts else np.zeros((0, 3))
    return np.array(offs, dtype=np.uint64), cat(m0s), cat(m1s)


def write_track_file(w: Window, path: str, max_num_iterations=10, function_tolerance=0.0,
                     gradient_tolerance=0.0, parameter_tolerance=0.0, use_loss_function=False,
                     fix_length=False, lowest_length=LOWEST_LENGTH, highest_length=HIGHEST_LENGTH):
    """Serialise a window (built with keep_tracks=True) as the reference's Frame/Feature track
    structure (sliding_window/Frame.hpp, Feature.hpp) for host/adapter_check: per frame the
    features in landmark order, each with its ray (x, y, 1) as projectPixelTo3dRay returns it
    (FeatureOperations.cpp:31-33), the index of its predecessor feature in the next older frame
    (_preFeature, -1 if none) and the number of older frames it reaches (_preFeatureCounter)."""
    import struct

    tr = w.meta["tracks"]
    lo, hi, rays = tr["lo"], tr["hi"], tr["rays"]
    N = w.N
    ids = [np.nonzero((lo <= i) & (hi >= i))[0] for i in range(N)]
    index = []
    for i in range(N):
        m = -np.ones(len(lo), dtype=np.int64)
        m[ids[i]] = np.arange(len(ids[i]))
        index.append(m)
    with open(path, "wb") as f:
        f.write(struct.pack("<5i", N, w.M, max_num_iterations, int(use_loss_function), int(fix_length)))
        f.write(struct.pack("<5d", function_tolerance, gradient_tolerance, parameter_tolerance,
                            lowest_length, highest_length))
        for i in range(N):
            f.write(np.ascontiguousarray(w.frames[i], dtype="<f8").tobytes())
            f.write(struct.pack("<i", len(ids[i])))
            for l in ids[i]:
                r = rays[i, l] / rays[i, l, 2]
                pre = int(index[i + 1][l]) if i + 1 < N and hi[l] >= i + 1 else -1
                cnt = int(min(hi[l], N - 1) - i)
                f.write(struct.pack("<3dii", r[0], r[1], r[2], pre, cnt))


# ---------------------------------------------------------------------------------------
# BASELINE.json configs.  seed = 1000 + config index (+ window id).
# ---------------------------------------------------------------------------------------
def config1(seed: int = 1001) -> Window:
    """N=10, M=5, 300 landmarks all visible: 3 000 bearing obs, 35 pairs, 10 500 blocks."""
    return make_window(10, 5, 300, seed, span="full")


def config2(seed: int = 1002, outlier_fraction: float = 0.0) -> Window:
    """N=30, M=5, 5 000 landmarks, spans of 20 frames: ~100k bearing obs."""
    return make_window(30, 5, 5000, seed, span=("sliding", 20), outlier_fraction=outlier_fraction)


def config3_window(window_id: int, seed: int = 1003) -> Window:
    """One of the 4 096 independent config-1-shaped windows."""
    return make_window(10, 5, 300, seed + window_id, span="full")


def config4(seed: int = 1004, n_active: int = 20000, outlier_fraction: float = 0.0) -> Window:
    """N=200, M=5, 985 pairs; `n_active` landmarks anchored at the newest frame with span
    U[2,200] (the other ~180k landmarks of the 200k scene never reach the newest M frames and
    therefore generate no residual blocks in the reference's formulation)."""
    return make_window(200, 5, n_active, seed, span=("anchored", 2, 200),
                       outlier_fraction=outlier_fraction)


def reference_defaults_window(seed: int = 7, n_features: int = 60) -> Window:
    """The reference's shipped defaults: numberToNote=3, numberToRefine=2, <=60 features
    (mvo/cfg/mvo.cfg:13,42-43)."""
    return make_window(3, 2, n_features, seed, span="full")

"""Shared helpers of the test-suite."""
import glob
import os

import numpy as np

from mvo_amd import scene

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                      if not os.path.basename(p).startswith(("scaler_", "epipolar_")))  # those: test_scaler_oracle.py, test_epipolar_oracle.py

FIXED10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
TIGHT = dict(max_num_iterations=400, function_tolerance=1e-16, gradient_tolerance=1e-14, parameter_tolerance=1e-14)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))  # allow_pickle stays False
    N, M = int(z["N"]), int(z["M"])
    w = scene.Window(N=N, M=M, frames=z["frames"].copy(), pair_offsets=z["pair_offsets"].copy(),
                     m0=z["m0"].copy(), m1=z["m1"].copy(), true_vec=np.zeros((N, 3)), true_scale=np.zeros(N))
    return w, z


def rel_inf(a, b):
    """max |a-b| relative to ||b||_inf (tolerances are relative to the vector norm)."""
    a, b = np.asarray(a), np.asarray(b)
    d = np.abs(b).max() if b.size else 0.0
    if d == 0.0:
        return float(np.abs(a).max()) if a.size else 0.0
    return float(np.abs(a - b).max() / d)


def refined(frames, M):
    return np.asarray(frames)[:M, 9:13]


def per_track_tolerance(R0, R1, R2, u1, v0, v1, v2):
    """The reference's formula takes sqrt(1 - d^2) of four dot products d of unit vectors: a last-bit
    difference in d (fma contraction, summation order) is amplified by 1 / (1 - d^2), up to 1e8 for
    nearly parallel rays.  Relative tolerance per track: 1e-14 / smallest radicand + 1e-12."""
    unit = lambda R, v: (v @ np.asarray(R).reshape(3, 3).T) / np.linalg.norm(v, axis=1, keepdims=True)
    m0, m1, m2 = unit(R0, v0), unit(R1, v1), unit(R2, v2)
    with np.errstate(invalid="ignore"):
        rad = np.stack([1 - (m2 @ u1) ** 2, 1 - np.sum(m1 * m0, 1) ** 2, 1 - np.sum(m2 * m1, 1) ** 2, 1 - (m0 @ u1) ** 2])
        return 1e-14 / np.abs(rad).min(axis=0) + 1e-12


def close_per_track(a, b, tol):
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= tol * np.abs(b)
    return bool(np.all(ok | (np.isnan(a) & np.isnan(b))))

"""Shared helpers of the test-suite."""
import glob
import os

import numpy as np

from mvo_amd import scene

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))

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

"""Golden vectors for the closed-form scale stage: an INDEPENDENT numpy restatement of
Scaler::stage's formula (mvo/src/pipeline/Scaler.cpp:34-67), vectorised, written without looking at
oracle/mvo_scaler_oracle.c's structure.  Run from the repo root:  python tests/golden/make_scaler_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import scaler_capi  # noqa: E402  (generator of the inputs only)


def scaler_numpy(t, lowest, highest):
    unit = lambda R, v: (v @ R.T) / np.linalg.norm(v, axis=1, keepdims=True)
    m0, m1, m2 = unit(t["R0"], t["v0"]), unit(t["R1"], t["v1"]), unit(t["R2"], t["v2"])
    u1 = t["u1"]
    s = lambda a: np.sqrt(1.0 - a ** 2)
    with np.errstate(invalid="ignore"):
        n0 = t["n1"] * s(m2 @ u1) * s(np.sum(m1 * m0, 1)) / (s(np.sum(m2 * m1, 1)) * s(m0 @ u1))
    srt = np.sort(n0)  # NaN last
    k = (len(n0) - 1) // 2
    med = (srt[k] + srt[k + 1]) / 2 if len(n0) % 2 == 0 else srt[k]
    return n0, med, float(np.clip(med, lowest, highest))


if __name__ == "__main__":
    for name, n, seed, lo, hi in (("scaler_60", 60, 3, 0.25, 2.0), ("scaler_61", 61, 4, 0.25, 2.0),
                                  ("scaler_5000", 5000, 5, 0.25, 2.0), ("scaler_clamp", 40, 6, 1.9, 2.0)):
        t = scaler_capi.synthetic_tracks(n, seed)
        n0, med, sc = scaler_numpy(t, lo, hi)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), lowest=lo, highest=hi,
                            per_track=n0, median=med, scale=sc, **{k: np.asarray(v) for k, v in t.items()})
        print(name, n, med, sc, t["true_n0"])

"""The Scaler restatement (oracle/mvo_scaler_oracle.c, following mvo/src/pipeline/Scaler.cpp:34-67)
against the golden vectors of an independent numpy restatement (tests/golden/make_scaler_golden.py).
The reference holds no fixtures for this stage: parity is pinned by formula."""
import glob
import os

import numpy as np
import pytest

import mvo_scaler_oracle_py as so
from _util import close_per_track, per_track_tolerance

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "scaler_*.npz")))


def load(path):
    return {k: v for k, v in np.load(path, allow_pickle=False).items()}


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(path):
    g = load(path)
    per, med, sc, cl, n = so.run(g["R0"], g["R1"], g["R2"], g["u1"], float(g["n1"]), float(g["lowest"]),
                                 float(g["highest"]), g["v0"], g["v1"], g["v2"])
    assert n == len(g["v0"]) and len(GOLDEN) == 4
    assert close_per_track(per, g["per_track"], per_track_tolerance(g["R0"], g["R1"], g["R2"], g["u1"], g["v0"], g["v1"], g["v2"]))
    assert abs(med - float(g["median"])) <= 1e-8 * abs(med)
    assert abs(sc - float(g["scale"])) <= 1e-8 * abs(sc)
    assert cl == (1 if med > g["highest"] else (-1 if med < g["lowest"] else 0))


def test_median_rule_even_odd_and_nan_last():
    I = np.eye(3)
    u1 = np.array([0.0, 0.0, 1.0])
    v = np.array([[0.1 * k + 0.05, 0.02 * k, 1.0] for k in range(6)])
    for n in (1, 2, 5, 6):
        per, med, sc, cl, got = so.run(I, I, I, u1, 1.0, 0.0, 1e9, v[:n], v[:n][::-1].copy(), v[:n] * 2.0)
        srt = np.sort(per)
        k = (n - 1) // 2
        assert med == ((srt[k] + srt[k + 1]) / 2 if n % 2 == 0 else srt[k])
    assert so.run(I, I, I, u1, 1.0, 0.0, 1e9, v[:0], v[:0], v[:0])[4] == 0

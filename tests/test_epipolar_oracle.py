"""The EpipolarGeometry restatement (oracle/mvo_epipolar_oracle.c, following
mvo/src/algorithms/EpipolarGeometry.cpp:11-176) against the golden vectors of an independent numpy
restatement (tests/golden/make_epipolar_golden.py: numpy.linalg.svd null direction, math.comb).
The reference holds no fixtures for this stage: parity is pinned by formula; the sign of a
direction is pinned by nothing (cv::SVD's convention is an implementation detail)."""
import glob
import os

import numpy as np
import pytest

import mvo_epipolar_oracle_py as eo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "epipolar_*.npz")))


def same_direction(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) <= tol


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(path):
    g = {k: v for k, v in np.load(path, allow_pickle=False).items()}
    assert len(GOLDEN) == 4
    r = eo.estimate(g["mhi"], g["mt"], g["samples"], float(g["ps"]), float(g["threshold"]))
    assert r["exhausted"] == 0
    assert r["iterations"] == int(g["g_iterations"]) and r["planned"] == int(g["g_planned"])
    assert r["best_hypothesis"] == int(g["g_best_hypothesis"])
    assert np.array_equal(r["hyp_inliers"], g["g_hyp_inliers"])
    assert np.allclose(r["hyp_prob"], g["g_hyp_prob"], rtol=1e-9, atol=1e-12)
    for a, b in zip(r["hyp_b"], g["g_hyp_b"]):
        assert same_direction(a, b, 1e-9)
    assert np.array_equal(r["inliers"], g["g_inliers"])
    assert same_direction(r["b"], g["g_b"], 1e-9)
    assert abs(r["best_probability"] - float(g["g_best_probability"])) <= 1e-9 * abs(r["best_probability"])


def test_too_few_features_and_exhaustion():
    z = np.array([[0.1, 0.2, 1.0], [0.3, 0.1, 1.0]])
    r = eo.estimate(z, z, np.zeros((4, 3), dtype=np.uint32), 0.99, 0.9)
    assert np.array_equal(r["b"], np.zeros(3)) and r["iterations"] == 0 and len(r["inliers"]) == 0
    g = {k: v for k, v in np.load(GOLDEN[0], allow_pickle=False).items()}
    r = eo.estimate(g["mhi"], g["mt"], g["samples"][:2], float(g["ps"]), float(g["threshold"]))
    assert r["exhausted"] == 1 and r["iterations"] == 2


def test_singular_sample_gives_zero_direction():
    m = np.array([[0.1, 0.2, 1.0], [0.3, -0.1, 1.0], [0.0, 0.4, 1.0]])
    assert np.array_equal(eo.baseline(m, m, [0, 1, 2]), np.zeros(3))      # a = 0 for identical rays: det = 0

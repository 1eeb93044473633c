"""GPU parity of the RANSAC baseline-direction stage (include/mvo_hip_epipolar.h) through the C ABI
against the CPU restatement of EpipolarGeometry::estimateBaseLine (oracle/mvo_epipolar_oracle.c) and
the numpy golden vectors.  Tolerances: inlier counts and index lists exact, summed probabilities
1e-9 relative, directions 1e-9 up to sign (pinned by nothing, see the header)."""
import glob
import os

import numpy as np
import pytest

import mvo_epipolar_oracle_py as eo
from mvo_amd import capi, epipolar_capi
from test_epipolar_oracle import same_direction

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "epipolar_*.npz")))


@pytest.fixture(scope="module")
def ep():
    e = epipolar_capi.Epipolar(0)
    yield e
    e.close()


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_golden(ep, path):
    g = {k: v for k, v in np.load(path, allow_pickle=False).items()}
    res, inl = ep.estimate(g["mhi"], g["mt"], g["samples"], float(g["ps"]), float(g["threshold"]))
    assert res.exhausted == 0 and res.iterations == int(g["g_iterations"])
    assert res.planned_iterations == int(g["g_planned"]) and res.best_hypothesis == int(g["g_best_hypothesis"])
    assert np.array_equal(inl, g["g_inliers"])
    assert same_direction(list(res.b), g["g_b"], 1e-9)
    assert abs(res.best_probability - float(g["g_best_probability"])) <= 1e-9 * abs(res.best_probability)
    k = res.iterations
    hb, hi, hp = ep.score(g["mhi"], g["mt"], g["samples"][:k], float(g["threshold"]))
    assert np.array_equal(hi, g["g_hyp_inliers"]) and np.allclose(hp, g["g_hyp_prob"], rtol=1e-9, atol=1e-12)
    for a, b in zip(hb, g["g_hyp_b"]):
        assert same_direction(a, b, 1e-9)


@pytest.mark.parametrize("n,outliers", [(3, 0.0), (7, 0.2), (60, 0.3), (500, 0.5), (2500, 0.2), (100_000, 0.3)])
def test_all_hypotheses_against_oracle(ep, n, outliers):
    t = epipolar_capi.synthetic_correspondences(n, seed=200 + n % 89, outlier_fraction=outliers, n_samples=300)
    thr = float(np.cos(np.deg2rad(4.0)))
    hb, hi, hp = ep.score(t["mhi"], t["mt"], t["samples"], thr)
    idx = np.arange(n, dtype=np.uint32)
    _, hio, hpo = eo.score(t["mhi"], t["mt"], t["samples"], thr)     # every hypothesis, every correspondence
    assert np.abs(hi.astype(np.int64) - hio.astype(np.int64)).max() <= 1   # a probability within rounding of the threshold
    same = hi == hio
    assert same.mean() > 0.98 and np.allclose(hp[same], hpo[same], rtol=1e-9, atol=1e-9)
    for k in range(0, 300, 7):
        b = eo.baseline(t["mt"], t["mhi"], t["samples"][k])
        assert same_direction(hb[k], b, 1e-9)
        p = np.array([eo.lib().mvo_epipolar_oracle_pinlier(t["mhi"][i].ctypes.data_as(eo.C.c_void_p),
                                                           t["mt"][i].ctypes.data_as(eo.C.c_void_p),
                                                           b.ctypes.data_as(eo.C.c_void_p)) for i in idx[:2000]])
        if n <= 2000:  # exact counts unless a probability sits within rounding of the threshold
            near = np.abs(p - thr) < 1e-12
            assert abs(int(hi[k]) - int((p > thr).sum())) <= int(near.sum())
            if not near.any():
                assert abs(hp[k] - p[p > thr].sum()) <= 1e-9 * max(1.0, abs(hp[k]))
    if n <= 2953:  # beyond that the reference's own binomial -> unsigned conversion overflows
        res, inl = ep.estimate(t["mhi"], t["mt"], t["samples"], 0.99, thr)
        r = eo.estimate(t["mhi"], t["mt"], t["samples"], 0.99, thr)
        assert res.iterations == r["iterations"] and res.planned_iterations == r["planned"]
        assert res.exhausted == r["exhausted"] and res.best_hypothesis == r["best_hypothesis"]
        assert np.array_equal(inl, r["inliers"]) and same_direction(list(res.b), r["b"], 1e-9)
        if outliers <= 0.2 and n >= 500:  # sanity of the data, not parity: the estimate points along the motion
            assert abs(np.dot(list(res.b), t["true_direction"])) > 0.95


def test_edge_cases(ep):
    t = epipolar_capi.synthetic_correspondences(40, seed=2)
    res, inl = ep.estimate(t["mhi"][:2], t["mt"][:2], np.zeros((4, 3), dtype=np.uint32))
    assert list(res.b) == [0.0, 0.0, 0.0] and res.iterations == 0 and len(inl) == 0   # EpipolarGeometry.cpp:44-48
    res, _ = ep.estimate(t["mhi"], t["mt"], t["samples"][:1])
    assert res.exhausted == 1 and res.iterations == 1
    with pytest.raises(capi.MvoHipError) as e:
        ep.estimate(t["mhi"], t["mt"], np.full((4, 3), 40, dtype=np.uint32))
    assert e.value.status == capi.ERR_INVALID_ARGUMENT
    same = np.tile(t["mhi"][:1], (5, 1))                      # identical rays: every A is singular, b = 0
    hb, hi, hp = ep.score(same, same, np.array([[0, 1, 2]], dtype=np.uint32), 0.9)
    assert np.array_equal(hb, np.zeros((1, 3))) and hi[0] == 5
    secs, ev = ep.last_timing()
    assert secs > 0 and ev == 5


def test_correspondences_on_the_threshold_are_decided_by_the_literal_expression(ep):
    """`pInlier > threshold` (EpipolarGeometry.cpp:96-100) feeds an integer: the inlier count, through it
    the adaptive iteration count (:104-106) and the winner (:113).  Correspondences built ON the
    threshold for hypothesis 0 (p = threshold to a few ulps, either side) must be counted exactly as
    the reference's own expression -- two square roots and a division, every operation rounded by
    itself -- decides, not as the scoring kernel's fused reciprocal square root would."""
    t = epipolar_capi.synthetic_correspondences(2000, seed=321, outlier_fraction=0.3, n_samples=64)
    thr = float(np.cos(np.deg2rad(4.0)))
    hb, _, _ = ep.score(t["mhi"], t["mt"], t["samples"], thr)
    b = hb[0] / np.linalg.norm(hb[0])
    e1 = np.cross(b, [0.3, -0.5, 0.8])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(b, e1)
    rng = np.random.default_rng(7)
    n_on = 4096
    al = rng.uniform(0, 2 * np.pi, n_on)
    th = np.arccos(thr)
    z1, z2 = rng.uniform(-0.5, 0.5, n_on), rng.uniform(-0.5, 0.5, n_on)
    unit = lambda v: v / np.linalg.norm(v, axis=1, keepdims=True)
    m1 = unit(np.cos(al)[:, None] * e1 + np.sin(al)[:, None] * e2 + z1[:, None] * b)
    m2 = unit(np.cos(al + th)[:, None] * e1 + np.sin(al + th)[:, None] * e2 + z2[:, None] * b)
    mhi = np.ascontiguousarray(np.concatenate([t["mhi"], m1]))
    mt = np.ascontiguousarray(np.concatenate([t["mt"], m2]))
    hb2, hi, hp = ep.score(mhi, mt, t["samples"], thr)   # the draws index the first 2000 only: same hypotheses
    assert np.array_equal(hb2, hb)

    def literal(bv):  # the reference's expression in its operation order, one rounding per operation
        d = lambda u, v: (u[:, 0] * v[:, 0] + u[:, 1] * v[:, 1]) + u[:, 2] * v[:, 2]
        bb = np.tile(bv, (len(mhi), 1))
        m1b, m2b, m12, n1, n2 = d(mhi, bb), d(mt, bb), d(mhi, mt), d(mhi, mhi), d(mt, mt)
        with np.errstate(invalid="ignore", divide="ignore"):
            return (m12 - m2b * m1b) / (np.sqrt(n1 - m1b * m1b) * np.sqrt(n2 - m2b * m2b))

    p0 = literal(hb[0])
    on = p0[2000:]
    assert np.abs(on - thr).max() < 1e-13                              # they really sit on the threshold ...
    assert 0.2 * n_on < np.count_nonzero(on > thr) < 0.8 * n_on        # ... on both sides of it
    for k in range(len(hb)):
        pk = literal(hb[k])
        assert hi[k] == np.count_nonzero(pk > thr), k                  # integer outputs: exact
        assert abs(hp[k] - pk[pk > thr].sum()) <= 1e-9 * max(1.0, hp[k])
    res, inl = ep.estimate(mhi, mt, t["samples"], 0.99, thr)
    pw = literal(hb[res.best_hypothesis])
    assert res.n_inliers == np.count_nonzero(pw > thr) == len(inl)

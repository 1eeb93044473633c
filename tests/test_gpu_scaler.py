"""GPU parity of the closed-form scale stage (include/mvo_hip_scaler.h) through the C ABI against the
CPU restatement of Scaler::stage (oracle/mvo_scaler_oracle.c) and the numpy golden vectors.
Tolerance: per-track scales 1e-14 / (smallest radicand 1 - d^2) + 1e-12 relative (_util.per_track_tolerance:
the formula's own cancellation amplifies last-bit differences by up to 1e8 for nearly parallel rays);
the median, a well-conditioned middle track, 1e-8."""
import glob
import os

import numpy as np
import pytest

import mvo_scaler_oracle_py as so
from _util import close_per_track, per_track_tolerance
from mvo_amd import capi, scaler_capi

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "scaler_*.npz")))


@pytest.fixture(scope="module")
def sc():
    s = scaler_capi.Scaler(0)
    yield s
    s.close()


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_golden(sc, path):
    g = {k: v for k, v in np.load(path, allow_pickle=False).items()}
    inp = scaler_capi.make_input(g["R0"], g["R1"], g["R2"], g["u1"], float(g["n1"]), float(g["lowest"]), float(g["highest"]))
    res, per = sc.scale(inp, g["v0"], g["v1"], g["v2"])
    assert res.n_tracks == len(g["v0"])
    assert close_per_track(per, g["per_track"], per_track_tolerance(g["R0"], g["R1"], g["R2"], g["u1"], g["v0"], g["v1"], g["v2"]))
    assert abs(res.median - float(g["median"])) <= 1e-8 * abs(res.median)
    assert abs(res.scale - float(g["scale"])) <= 1e-8 * abs(res.scale)


@pytest.mark.parametrize("n", [1, 2, 3, 60, 2047, 2048, 2049, 2050, 100_001, 1_000_000])
def test_against_oracle_all_sizes(sc, n):
    t = scaler_capi.synthetic_tracks(n, seed=100 + n % 97)
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0)
    res, per = sc.scale(inp, t["v0"], t["v1"], t["v2"])
    pero, med, scl, cl, got = so.run(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0, t["v0"], t["v1"], t["v2"])
    assert res.n_tracks == got == n
    assert close_per_track(per, pero, per_track_tolerance(t["R0"], t["R1"], t["R2"], t["u1"], t["v0"], t["v1"], t["v2"]))
    # the device's median must be exactly the middle order statistic(s) of the device's own scales
    srt = np.sort(per)
    k = (n - 1) // 2
    assert res.median == ((srt[k] + srt[k + 1]) / 2 if n % 2 == 0 else srt[k])
    assert abs(res.median - med) <= 1e-8 * abs(med) and res.clamped == cl
    assert abs(res.scale - scl) <= 1e-8 * abs(scl)


def test_duplicates_nans_and_clamping(sc):
    t = scaler_capi.synthetic_tracks(5000, seed=9)
    v0 = t["v0"].copy()
    v0[:3000] = v0[0]                      # many equal keys around the median
    t["v1"][:3000] = t["v1"][0]
    t["v2"][:3000] = t["v2"][0]
    v0[4000:4100, 2] = np.nan              # NaN scales sort last
    for lo, hi in ((0.25, 2.0), (5.0, 6.0), (1e-6, 1e-5)):
        inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], lo, hi)
        res, per = sc.scale(inp, v0, t["v1"], t["v2"])
        pero, med, scl, cl, _ = so.run(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], lo, hi, v0, t["v1"], t["v2"])
        assert np.isnan(per[4000:4100]).all()
        assert close_per_track(per, pero, per_track_tolerance(t["R0"], t["R1"], t["R2"], t["u1"], v0, t["v1"], t["v2"]))
        assert abs(res.median - med) <= 1e-8 * abs(med) and res.clamped == cl and abs(res.scale - scl) <= 1e-8 * abs(scl)


@pytest.mark.parametrize("n", [30_001, 200_000])
@pytest.mark.parametrize("order", ["largest_first", "smallest_first", "two_binades"])
def test_select_when_the_first_tracks_are_unlike_the_rest(sc, n, order):
    """k_scaler_tracks guesses the median's top digit from the first 64 tracks to save one pass over
    the keys.  Here the guess is wrong on purpose (the first tracks hold the extreme scales) or the
    scales straddle two binades: the select must take its separate pass and return the same median."""
    t = scaler_capi.synthetic_tracks(n, seed=700 + n % 13, noise_sigma=2e-2)
    pero = so.run(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0, t["v0"], t["v1"], t["v2"])[0]
    key = np.where(np.isnan(pero), np.inf, pero)
    if order == "two_binades":
        perm = np.argsort(np.abs(key - 1.0), kind="stable")  # 0.99.., 1.00.. alternate around 1.0
        n1 = t["n1"] / np.median(key)                        # median scale ~ 1.0: digit 1 splits it
    else:
        perm = np.argsort(key if order == "smallest_first" else -key, kind="stable")
        n1 = t["n1"]
    v0, v1, v2 = (np.ascontiguousarray(t[k][perm]) for k in ("v0", "v1", "v2"))
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], n1, 0.25, 2.0)
    res, per = sc.scale(inp, v0, v1, v2)
    pero2, med, scl, cl, _ = so.run(t["R0"], t["R1"], t["R2"], t["u1"], n1, 0.25, 2.0, v0, v1, v2)
    if order != "two_binades":  # the premise: the first 64 scales sit in another binade than the median
        first = np.median(pero2[:64])
        assert not (0.5 <= first / med < 2.0) or np.floor(np.log2(first)) != np.floor(np.log2(med))
    srt = np.sort(per)
    k = (n - 1) // 2
    assert res.median == ((srt[k] + srt[k + 1]) / 2 if n % 2 == 0 else srt[k])
    assert abs(res.median - med) <= 1e-8 * abs(med) and res.clamped == cl


def test_empty_window_and_split_form(sc):
    t = scaler_capi.synthetic_tracks(10, seed=1)
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"])
    res, _ = sc.scale(inp, t["v0"][:0], t["v1"][:0], t["v2"][:0])
    assert res.n_tracks == 0 and res.scale == 0.0          # Scaler.cpp:25-26: nothing happens
    sc.upload(t["v0"], t["v1"], t["v2"])
    a, _ = sc.run(inp)
    b, per = sc.run(inp, per_track=True)
    assert a.median == b.median and len(per) == 10
    secs, nbytes = sc.last_timing()
    assert secs > 0 and nbytes == 80 * 10
    with pytest.raises(capi.MvoHipError):
        sc._lib.mvo_hip_scaler_run.restype = None or sc._lib.mvo_hip_scaler_run.restype
        rc = sc._lib.mvo_hip_scaler_run(sc._h, None, None, None)
        raise capi.MvoHipError(rc, "mvo_hip_scaler_run") if rc else AssertionError("expected an error status")

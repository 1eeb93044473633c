"""Run-to-run reproducibility through the C ABI: no kernel here uses atomics on floating point data
or an unordered reduction, so the same call on the same handle must return the same BITS every
time -- a race between streams, a missing barrier or a read of stale scratch shows up as a
difference.  Every entry point of the five headers that computes something is repeated."""
import numpy as np
import pytest

from mvo_amd import ba_capi, ba_scene, capi, epipolar_capi, scaler_capi, scene, tracks_capi

pytestmark = pytest.mark.gpu
REPS = 6


def _same(first, again):
    if isinstance(first, np.ndarray):
        return np.array_equal(first, again, equal_nan=True)
    return first == again


@pytest.mark.parametrize("use_loss", [False, True])
@pytest.mark.parametrize("shape", ["defaults", "config1", "batch64", "ragged"])
def test_refine_is_bit_reproducible(solver, shape, use_loss):
    if shape == "defaults":  # N = 3, M = 2: inline evaluation, M < 5
        ws = [scene.reference_defaults_window()]
    elif shape == "config1":  # tiled passes, one window
        ws = [scene.config1()]
    elif shape == "batch64":  # many windows, one workgroup each
        ws = [scene.config3_window(i) for i in range(64)]
    else:
        ws = [scene.make_window(8, 3, 500, seed=77, span=("sliding", 4))]
    opt = capi.make_options(max_num_iterations=12, function_tolerance=0.0, gradient_tolerance=0.0,
                            parameter_tolerance=0.0, use_loss_function=use_loss)
    first = None
    for rep in range(REPS):
        solver.upload(ws, opt)
        solver.solve()
        frames, summ = solver.download()
        got = (np.array(frames), [(s.iterations, s.termination, s.initial_cost, s.final_cost) for s in summ])
        if first is None:
            first = got
        assert _same(first[0], got[0]) and first[1] == got[1], f"repetition {rep} differs"


@pytest.mark.parametrize("loss", [ba_capi.LOSS_NONE, ba_capi.LOSS_CAUCHY])
def test_ba_is_bit_reproducible(loss):
    sc = ba_scene.make_ba_scene(12, 9, 400, seed=31, span=7, outlier_fraction=0.05 if loss else 0.0)
    opt = ba_capi.make_options(max_num_iterations=6, loss=loss, loss_scale=0.004, function_tolerance=0.0,
                               gradient_tolerance=0.0, parameter_tolerance=0.0)
    with ba_capi.BaSolver(0) as ba:
        first = None
        for rep in range(REPS):
            poses, X, s = ba.refine(sc, opt)
            got = (poses, X, (s.iterations, s.termination, s.initial_cost, s.final_cost))
            if first is None:
                first = got
            assert _same(first[0], got[0]) and _same(first[1], got[1]) and first[2] == got[2], f"repetition {rep} differs"


def test_scaler_and_epipolar_are_bit_reproducible():
    t = scaler_capi.synthetic_tracks(20011, seed=5)
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0)
    c = epipolar_capi.synthetic_correspondences(3001, seed=6, n_samples=512)
    with scaler_capi.Scaler(0) as sc, epipolar_capi.Epipolar(0) as ep:
        first = None
        for rep in range(REPS):
            res, per = sc.scale(inp, t["v0"], t["v1"], t["v2"])
            hb, hi, hp = ep.score(c["mhi"], c["mt"], c["samples"])
            est, inl = ep.estimate(c["mhi"], c["mt"], c["samples"])
            got = (per, (res.median, res.n_tracks), hb, hi, hp, inl,
                   (tuple(est.b), est.best_probability, est.n_inliers, est.iterations, est.best_hypothesis))
            if first is None:
                first = got
            assert all(_same(a, b) for a, b in zip(first, got)), f"repetition {rep} differs"


def test_track_tables_are_bit_reproducible(solver):
    w = scene.make_window(12, 5, 1500, seed=4300, span=("anchored", 2, 12), keep_tracks=True)
    frames = scene.track_frames(w)
    opt = capi.make_options(max_num_iterations=8, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    first = None
    for rep in range(REPS):  # a NEW table every time: creation, pushes and the first gather race nothing
        with tracks_capi.TrackTable(solver, max_frames=13, max_features=max(len(p) for _, p in frames)) as table:
            for rays, pre in frames:
                table.push(rays, pre)
            got_frames, gs = table.refine(w.frames, 5, opt)
            offs, m0, m1 = table.gather(5, 12)
        got = (np.array(got_frames), offs, m0, m1, gs.final_cost)
        if first is None:
            first = got
        assert all(_same(a, b) for a, b in zip(first, got)), f"repetition {rep} differs"

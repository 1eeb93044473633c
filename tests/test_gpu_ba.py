"""GPU parity of the explicit-landmark window bundle adjustment (include/mvo_hip_ba.h) against the
self-oracle oracle/mvo_ba_oracle.c, through the C ABI.  PARITY UNPINNED: no reference counterpart
(SURVEY.md 8f rank 1).  Tolerances: residuals, reduced camera system and gradient 1e-9 relative to
the largest entry; refined poses/landmarks 1e-6 after a fixed iteration count."""
import numpy as np
import pytest

import mvo_ba_oracle_py as bo
from mvo_amd import ba_capi, ba_scene, capi

pytestmark = pytest.mark.gpu

FIXED = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


@pytest.fixture(scope="module")
def ba():
    s = ba_capi.BaSolver(0)
    yield s
    s.close()


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


SCENES = {
    "tiny_all_visible": lambda: ba_scene.make_ba_scene(6, 4, 40, seed=5),
    "one_free_frame": lambda: ba_scene.make_ba_scene(5, 1, 30, seed=6),
    "config1": ba_scene.ba_config1,
    "spans_28_free": lambda: ba_scene.make_ba_scene(30, 28, 600, seed=8, span=12),
    "mostly_fixed": lambda: ba_scene.make_ba_scene(40, 6, 500, seed=9, span=10),
    "long_tracks": lambda: ba_scene.make_ba_scene(100, 5, 64, seed=10),  # 100 observations per landmark
    # more frames than the kernels stage in LDS: k_ba_step reads poses through L2 from 257 frames on,
    # k_ba_schur / k_ba_fixed from 513 on
    "frames_300": lambda: ba_scene.make_ba_scene(300, 7, 900, seed=12, span=30),
    "frames_600": lambda: ba_scene.make_ba_scene(600, 9, 1500, seed=13, span=25),
}


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("loss", [0, 1, 2])
def test_residuals_cost_and_reduced_system(ba, name, loss):
    sc = SCENES[name]()
    kw = dict(loss=loss, loss_scale=0.002)
    ba.upload(sc, ba_capi.make_options(**kw))
    r, cost = ba.evaluate()
    ro, costo = bo.residuals(sc, bo.make_options(**kw))
    assert _rel(r, ro) < 1e-9 and abs(cost - costo) <= 1e-11 * costo
    for radius in (1e4, 3.7):
        S, rhs, gc, gm, c = ba.reduced(radius)
        So, rhso, go, co = bo.reduced(sc, bo.make_options(**kw), radius)
        assert abs(c - co) <= 1e-11 * co
        assert _rel(gc, go[:6 * sc.M]) < 1e-9
        assert abs(gm - np.abs(go).max()) <= 1e-9 * np.abs(go).max()
        assert _rel(S, So) < 1e-9 and _rel(rhs, rhso) < 1e-9
        assert np.array_equal(S, S.T)


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("loss", [0, 1])
def test_fixed_iterations_match_oracle(ba, name, loss):
    sc = SCENES[name]()
    kw = dict(max_num_iterations=8, loss=loss, loss_scale=0.003, **FIXED)
    ba.set_trace(64)
    poses, X, s = ba.refine(sc, ba_capi.make_options(**kw))
    po, Xo, so, tro = bo.refine(sc, bo.make_options(**kw))
    tr = ba.trace()
    ba.set_trace(0)
    assert s.iterations == so.iterations == 8 and s.termination == so.termination == 1
    assert len(tr) == len(tro)
    for a, b in zip(tr, tro):  # once the cost moves by rounding noise only, accept/reject is noise too
        if abs(b.cost_change) > 1e-9 * b.cost:
            assert a.step_is_successful == b.step_is_successful
            assert abs(a.cost - b.cost) <= 1e-8 * b.cost
            assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-5 * b.trust_region_radius
    assert abs(s.initial_cost - so.initial_cost) <= 1e-11 * so.initial_cost
    assert abs(s.final_cost - so.final_cost) <= 1e-7 * so.final_cost
    assert np.abs(poses - po).max() < 1e-6 and _rel(X, Xo) < 1e-6
    assert np.array_equal(poses[sc.M:], sc.poses[sc.M:])


def test_default_tolerances_converge_like_oracle(ba):
    sc = ba_scene.make_ba_scene(12, 8, 300, seed=21, span=8)
    poses, X, s = ba.refine(sc, ba_capi.make_options(max_num_iterations=50))
    po, Xo, so, _ = bo.refine(sc, bo.make_options(max_num_iterations=50))
    assert s.termination == so.termination == 0 and s.iterations == so.iterations
    assert np.abs(poses - po).max() < 1e-6 and _rel(X, Xo) < 1e-6


def test_noise_free_scene_is_recovered(ba):
    sc = ba_scene.make_ba_scene(8, 6, 120, seed=11, noise_sigma=0.0)
    poses, X, s = ba.refine(sc, ba_capi.make_options(max_num_iterations=30, function_tolerance=1e-16,
                                                     gradient_tolerance=1e-14, parameter_tolerance=1e-14))
    assert s.final_cost < 1e-20 < s.initial_cost
    assert np.abs(poses - sc.true_poses).max() < 1e-8 and np.abs(X - sc.true_X).max() < 1e-6


def test_split_phase_restarts_from_uploaded_point(ba):
    sc = ba_scene.ba_config1()
    opt = ba_capi.make_options(max_num_iterations=5, **FIXED)
    ba.upload(sc, opt)
    ba.solve()
    a = ba.download()
    ba.solve()
    b = ba.download()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])  # no atomics: bitwise reproducible
    assert a[2].iterations == b[2].iterations == 5


def test_error_behaviour(ba):
    sc = ba_scene.make_ba_scene(6, 4, 20, seed=3)
    with pytest.raises(capi.MvoHipError) as e:
        ba.upload(ba_scene.make_ba_scene(40, 29, 20, seed=3), ba_capi.make_options())
    assert e.value.status == capi.ERR_UNSUPPORTED
    bad = ba_scene.make_ba_scene(6, 4, 20, seed=3)
    bad.obs_frame = bad.obs_frame.copy()
    bad.obs_frame[0] = 77
    with pytest.raises(capi.MvoHipError) as e:
        ba.upload(bad, ba_capi.make_options())
    assert e.value.status == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.MvoHipError) as e:
        ba.upload(sc, ba_capi.make_options(loss=1, loss_scale=0.0))
    assert e.value.status == capi.ERR_INVALID_ARGUMENT
    ba.upload(sc, ba_capi.make_options(max_num_iterations=0))
    ba.solve()
    poses, X, s = ba.download()
    assert s.iterations == 0 and s.termination == 1 and np.array_equal(X, sc.X)


def test_non_finite_observation_ends_as_failure_like_the_oracle(ba):
    """A NaN bearing: no valid step exists; both sides give up after five invalid steps (FAILURE) and
    hand the poses and landmarks back untouched and finite."""
    sc = ba_scene.make_ba_scene(8, 5, 120, seed=3, span=6)
    sc.obs_ray[17, 1] = np.nan
    kw = dict(max_num_iterations=8, **FIXED)
    poses, X, s = ba.refine(sc, ba_capi.make_options(**kw))
    po, Xo, so, _ = bo.refine(sc, bo.make_options(**kw))
    assert s.termination == so.termination == 2 and s.iterations == so.iterations
    assert np.array_equal(poses, sc.poses) and np.array_equal(po, sc.poses)
    assert np.array_equal(X, sc.X) and np.array_equal(Xo, sc.X)
    good = ba_scene.make_ba_scene(8, 5, 120, seed=3, span=6)  # and the handle still works
    poses, X, s = ba.refine(good, ba_capi.make_options(**kw))
    po, Xo, so, _ = bo.refine(good, bo.make_options(**kw))
    assert s.termination == so.termination and np.abs(poses - po).max() < 1e-6


@pytest.mark.parametrize("name", ["one_free_frame", "tiny_all_visible", "config1", "spans_28_free", "mostly_fixed"])
def test_mfma_factorisation_gives_the_same_solve(ba, name):
    """k_ba_solve_mfma (MVO_HIP_BA_SOLVE_MFMA=1, read at upload: the block L D L^T with its trailing
    update on v_mfma_f64_16x16x4 -- measured slower than the thread-per-block kernel, kept for the
    record) against the default kernel and the self-oracle: same accept/reject sequence, poses 1e-9."""
    import os
    sc = SCENES[name]()
    kw = dict(max_num_iterations=8, loss=1, loss_scale=0.003, **FIXED)
    poses, X, s = ba.refine(sc, ba_capi.make_options(**kw))
    os.environ["MVO_HIP_BA_SOLVE_MFMA"] = "1"
    try:
        ba.set_trace(64)
        poses2, X2, s2 = ba.refine(sc, ba_capi.make_options(**kw))
        tr = ba.trace()
        ba.set_trace(0)
    finally:
        del os.environ["MVO_HIP_BA_SOLVE_MFMA"]
    po, Xo, so, tro = bo.refine(sc, bo.make_options(**kw))
    assert s2.iterations == s.iterations == so.iterations and s2.termination == s.termination
    assert [t.step_is_successful for t in tr] == [t.step_is_successful for t in tro] or \
        all(abs(b.cost_change) <= 1e-9 * b.cost for a, b in zip(tr, tro) if a.step_is_successful != b.step_is_successful)
    assert abs(s2.final_cost - s.final_cost) <= 1e-10 * s.final_cost
    assert np.abs(poses2 - poses).max() < 1e-9 and _rel(X2, X) < 1e-9
    assert np.abs(poses2 - po).max() < 1e-6 and _rel(X2, Xo) < 1e-6

"""CPU tests of the oracle (oracle/mvo_oracle.c) against the committed golden vectors, against
derivative checks, and against the reference's documented quirks.  No GPU needed.

Reference = /root/reference/mvo/src/algorithms/IterativeRefinement.cpp ("IR.cpp").
"""
import numpy as np
import pytest

from _util import FIXED10, GOLDEN_CASES, TIGHT, load_golden, refined, rel_inf
from mvo_amd import scene


# ------------------------------------------------------------------ scalar pieces
def test_pair_enumeration_matches_reference_loops(oracle):
    # IR.cpp:116,134: for a < M: for b in [a, N-1)
    for M, N in [(1, 2), (2, 3), (3, 6), (5, 10), (5, 30), (5, 200)]:
        pairs = scene.pair_list(M, N)
        assert oracle.lib().mvo_oracle_num_pairs(M, N) == len(pairs) == M * (N - 1) - M * (M - 1) // 2
    assert oracle.lib().mvo_oracle_num_pairs(5, 10) == 35
    assert oracle.lib().mvo_oracle_num_pairs(5, 30) == 135
    assert oracle.lib().mvo_oracle_num_pairs(5, 200) == 985
    assert oracle.lib().mvo_oracle_num_pairs(3, 3) == -1  # IR.cpp:29 assert(numberToNote > numberToRefine)
    assert oracle.lib().mvo_oracle_num_pairs(0, 3) == -1  # IR.cpp:30


def test_scale_roundtrip_and_bounds(oracle):
    # IR.cpp:298-309 / :348-368
    for n in [0.2500001, 0.3, 0.7, 1.0, 1.5, 1.9999]:
        t = oracle.reverse_scale(n)
        assert abs(oracle.scale(t) - n) <= 4e-16 * max(1.0, abs(t)) + 1e-15
    # clamping recursion: lengths outside (MIN, MAX) are pulled just inside
    eps = np.finfo(float).eps
    assert oracle.reverse_scale(0.1) == oracle.reverse_scale(0.25 + 0.1 * eps + eps)
    assert oracle.reverse_scale(0.25) == oracle.reverse_scale(0.25 + 0.25 * eps + eps)
    assert oracle.reverse_scale(5.0) == oracle.reverse_scale(2.0 - (5.0 * eps + eps))
    assert np.isfinite(oracle.reverse_scale(0.0)) and np.isfinite(oracle.reverse_scale(2.0))
    # saturation branch: exp(-t) = inf -> MIN_LEN with zero derivative (IR.cpp:302-304)
    assert oracle.scale(-800.0) == 0.25
    assert oracle.scale_derivative(-800.0) == 0.0
    assert oracle.scale(800.0) == 2.0
    # derivative against central differences
    for t in [-3.0, -0.5, 0.0, 0.7, 4.0]:
        h = 1e-6
        fd = (oracle.scale(t + h) - oracle.scale(t - h)) / (2 * h)
        assert abs(oracle.scale_derivative(t) - fd) < 1e-9


def test_plus_is_norm_preserving_and_has_documented_jacobian(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        x = rng.normal(size=3)
        x /= np.linalg.norm(x)
        d = rng.normal(scale=0.3, size=2)
        y = oracle.plus(x, d)
        assert abs(np.linalg.norm(y) - 1.0) < 1e-14
        assert np.allclose(oracle.plus(x, [0.0, 0.0]), x, atol=0, rtol=0)
        # d Plus / d delta at 0 = [[-2x1, 2x2], [2x0, 0], [0, -2x0]]  
        h = 1e-7
        J = np.stack([(oracle.plus(x, [h, 0]) - oracle.plus(x, [-h, 0])) / (2 * h),
                      (oracle.plus(x, [0, h]) - oracle.plus(x, [0, -h])) / (2 * h)], axis=1)
        Jdoc = np.array([[-2 * x[1], 2 * x[2]], [2 * x[0], 0.0], [0.0, -2 * x[0]]])
        assert np.abs(J - Jdoc).max() < 1e-8


def test_norm_features(oracle):
    import ctypes as C

    rays = np.array([[3.0, 4.0, 1.0], [0.1, -0.2, 1.0]])
    want = rays / np.linalg.norm(rays, axis=1, keepdims=True)
    oracle.lib().mvo_oracle_norm_features(rays.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(2))
    assert np.abs(rays - want).max() < 1e-15


# ------------------------------------------------------------------ residual / Jacobian
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_residual_jacobian_moments_match_golden(oracle, case):
    w, z = load_golden(case)
    r, J, cost = oracle.evaluate(w)
    assert rel_inf(r, z["r0"]) < 1e-12
    assert rel_inf(J, z["J0"]) < 1e-11  # golden J is complex-step of the reference formula
    assert abs(cost - float(z["cost0"])) < 1e-12 * float(z["cost0"])
    S = oracle.pair_moments(w)
    assert rel_inf(S, z["S"]) < 1e-13


@pytest.mark.parametrize("M,N,L", [(1, 2, 30), (2, 3, 60), (3, 6, 50), (5, 10, 40), (5, 12, 30)])
def test_analytic_jacobian_equals_autodiff(oracle, M, N, L):
    w = scene.make_window(N, M, L, seed=17 * M + N)
    r, J, c = oracle.evaluate(w)
    r2, J2, c2 = oracle.evaluate(w, autodiff=True)
    assert rel_inf(r, r2) < 1e-14
    assert rel_inf(J, J2) < 1e-13
    assert abs(c - c2) <= 1e-14 * c


def test_jacobian_central_differences(oracle):
    w = scene.make_window(6, 3, 25, seed=5)
    r0, J, _ = oracle.evaluate(w)
    h = 1e-6
    for k in range(w.M):
        for j in range(3):
            wp, wm = [scene.Window(**{**w.__dict__, "frames": w.frames.copy()}) for _ in range(2)]
            for ww, sgn in ((wp, 1.0), (wm, -1.0)):
                if j < 2:
                    d = [sgn * h if j == 0 else 0.0, sgn * h if j == 1 else 0.0]
                    ww.frames[k, 9:12] = oracle.plus(w.frames[k, 9:12], d)
                else:
                    t = oracle.reverse_scale(w.frames[k, 12]) + sgn * h
                    ww.frames[k, 12] = oracle.scale(t)
            rp, _, _ = oracle.evaluate(wp)
            rm, _, _ = oracle.evaluate(wm)
            fd = (rp - rm) / (2 * h)
            assert np.abs(fd - J[:, 3 * k + j]).max() < 2e-8 * max(1.0, np.abs(J).max())


def test_offset_quirk_only_frame_b_is_added(oracle):
    """IR.cpp:137-144: a fresh zero offset per pair, so for b > M the fixed baselines of frames
    M..b-1 are NOT part of the summed baseline.  Changing frame M's baseline must therefore leave
    the residuals of pairs with b > M untouched, and change those with b == M."""
    w = scene.make_window(9, 3, 20, seed=9)
    r0, _, _ = oracle.evaluate(w)
    w2 = scene.Window(**{**w.__dict__, "frames": w.frames.copy()})
    w2.frames[w.M, 12] *= 1.37  # scale of the first fixed frame
    r1, _, _ = oracle.evaluate(w2)
    for p, (a, b) in enumerate(scene.pair_list(w.M, w.N)):
        lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
        changed = np.abs(r1[lo:hi] - r0[lo:hi]).max() > 0
        assert changed == (b == w.M), (a, b)


def test_residual_does_not_depend_on_scale_for_single_block_pairs(oracle):
    """CostFunction1 (IR.cpp:466-471): u normalises the single s*v, so r is scale free."""
    w = scene.make_window(4, 2, 30, seed=2)
    _, J, _ = oracle.evaluate(w)
    pairs = scene.pair_list(w.M, w.N)
    for p, (a, b) in enumerate(pairs):
        lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
        if a == b and b < w.M:  # exactly one free block and no offset
            assert np.abs(J[lo:hi, 3 * a + 2]).max() < 1e-15


# ------------------------------------------------------------------ solver
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_converged_minimiser_matches_scipy(oracle, case):
    w, z = load_golden(case)
    opt = oracle.make_options(use_loss_function=bool(z["use_loss"]), fix_length=bool(z["fix_length"]), **TIGHT)
    f, s, tr = oracle.refine(w, opt)
    # same minimum of the same cost: final cost to 1e-9 relative, parameters to 1e-6
    assert abs(s.final_cost - float(z["cost_star"])) <= 1e-9 * float(z["cost_star"])
    assert np.abs(refined(f, w.M)[:, :3] - z["vec_star"]).max() < 1e-6
    assert np.abs(refined(f, w.M)[:, 3] - z["scale_star"]).max() < 1e-6 * max(1.0, z["scale_star"].max())


def test_lm_trace_follows_ceres_rules(oracle):
    w = scene.config1()
    f, s, tr = oracle.refine(w, oracle.make_options(**FIXED10))
    assert s.termination == 1 and s.reason == 1 and s.iterations == 10 and len(tr) == 11
    assert tr[0]["trust_region_radius"] == 1e4  # initial_trust_region_radius
    assert s.num_successful_steps + s.num_unsuccessful_steps == 11
    radius, factor = 1e4, 2.0
    for prev, it in zip(tr[:-1], tr[1:]):
        if it["step_is_successful"]:
            q = it["relative_decrease"]
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2 * q - 1) ** 3))
            factor = 2.0
            assert q > 1e-3
        else:
            radius /= factor
            factor *= 2.0
            assert it["relative_decrease"] <= 1e-3
        assert abs(it["trust_region_radius"] - radius) <= 1e-12 * radius
    assert abs(s.final_cost - min(t["cost"] for t in tr)) < 1e-18
    assert s.initial_cost == tr[0]["cost"]


def test_zero_iterations_leaves_input_untouched(oracle):
    w = scene.make_window(5, 2, 30, seed=4)
    f, s, tr = oracle.refine(w, oracle.make_options(max_num_iterations=0))
    assert s.iterations == 0 and s.reason == 1
    assert np.abs(refined(f, w.M)[:, :3] - refined(w.frames, w.M)[:, :3]).max() == 0.0
    assert np.abs(refined(f, w.M)[:, 3] - refined(w.frames, w.M)[:, 3]).max() < 1e-15  # s(reverse(s))


def test_default_tolerances_converge(oracle):
    w = scene.make_window(7, 5, 200, seed=5)
    f, s, tr = oracle.refine(w, oracle.make_options())
    assert s.termination == 0 and s.final_cost < s.initial_cost
    # direction error shrinks from the 3 degree initial perturbation
    d0 = np.sum(refined(w.frames, w.M)[:, :3] * w.true_vec[: w.M], axis=1)
    d1 = np.sum(refined(f, w.M)[:, :3] * w.true_vec[: w.M], axis=1)
    assert np.all(d1 > d0)


def test_fix_length_keeps_scales(oracle):
    w = scene.make_window(6, 3, 60, seed=8)
    f, s, _ = oracle.refine(w, oracle.make_options(fix_length=True, **FIXED10))
    assert np.abs(refined(f, w.M)[:, 3] - refined(w.frames, w.M)[:, 3]).max() < 1e-15
    assert np.abs(refined(f, w.M)[:, :3] - refined(w.frames, w.M)[:, :3]).max() > 1e-4


def test_empty_window_terminates(oracle):
    w = scene.make_window(4, 2, 10, seed=1)
    w.m0 = w.m0[:0]
    w.m1 = w.m1[:0]
    w.pair_offsets = np.zeros_like(w.pair_offsets)
    f, s, _ = oracle.refine(w, oracle.make_options())
    assert s.initial_cost == 0.0 and s.termination == 0  # gradient tolerance at iteration 0
    assert np.abs(refined(f, w.M)[:, :3] - refined(w.frames, w.M)[:, :3]).max() == 0.0


def test_rejects_bad_shapes(oracle):
    w = scene.make_window(4, 2, 10, seed=1)
    w.M, w.N = 6, 8  # the reference asserts false beyond 5 refined frames (IR.cpp:450)
    with pytest.raises(RuntimeError):
        oracle.refine(w)


@pytest.mark.parametrize("use_loss", [False, True])
def test_threaded_timing_path_agrees_with_the_serial_path(oracle, use_loss):
    """set_threads(n > 1) exists for bench.py's cpu_baseline legs only (pairs evaluated in parallel,
    QR by row ranges): same algorithm, different summation order -- results must agree with the
    serial parity path to rounding, and the default must stay 1."""
    assert oracle.get_threads() == 1
    w = scene.make_window(12, 5, 1500, 77, span=("sliding", 8), outlier_fraction=0.1 if use_loss else 0.0)
    opt = oracle.make_options(use_loss_function=use_loss, **FIXED10)
    f1, s1, _ = oracle.refine(w, opt)
    r1, j1, c1 = oracle.evaluate(w, opt)
    oracle.set_threads(3)
    try:
        f3, s3, _ = oracle.refine(w, opt)
        r3, j3, c3 = oracle.evaluate(w, opt)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(r1, r3) and np.array_equal(j1, j3)  # per-row values do not depend on threads
    assert abs(c1 - c3) <= 1e-13 * abs(c1)
    assert np.abs(refined(f1, w.M) - refined(f3, w.M)).max() < 1e-9
    assert s1.iterations == s3.iterations and abs(s1.final_cost - s3.final_cost) <= 1e-11 * abs(s1.final_cost)

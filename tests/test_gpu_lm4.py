"""k_lm4 (four windows per wavefront, csrc/mvo_lm4.h) against k_lm<64> (one window per wavefront) and
against the oracle: the batched solver of mvo_hip_refine_batch for windows with at most 64 frame pairs.
The LM step of k_lm4 is the branch-free restatement of lm_decide / lm_step: same arithmetic, so the
two kernels must take the same decisions and agree to rounding, whatever the windows of a wavefront do
(different accept / reject sequences, different termination evaluations, rows without a window)."""
import numpy as np
import pytest

from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu

FIXED10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


def _batch(n, N=10, M=5, feats=120, seed0=300, **kw):
    return [scene.make_window(N, M, feats, seed=seed0 + i, **kw) for i in range(n)]


def _solve(solver, ws, **opt):
    return solver.refine_batch(ws, capi.make_options(**opt))


@pytest.mark.parametrize("n", [4, 5, 7, 16, 33])
@pytest.mark.parametrize("opts", [dict(**FIXED10), dict(max_num_iterations=50), dict(fix_length=True, **FIXED10),
                                  dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                       parameter_tolerance=1e-8)])
def test_four_per_wave_equals_one_per_wave(solver, n, opts):
    ws = _batch(n)
    f4, s4 = _solve(solver, ws, assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_FOUR_PER_WAVE, **opts)
    f1, s1 = _solve(solver, ws, assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_ONE_PER_WAVE, **opts)
    for a, b in zip(s4, s1):
        ka = (a.termination, a.reason, a.iterations, a.num_successful_steps, a.num_unsuccessful_steps)
        kb = (b.termination, b.reason, b.iterations, b.num_successful_steps, b.num_unsuccessful_steps)
        if ka != kb:
            # function tolerance 0: the solve ends at the first evaluation that repeats the cost BIT FOR BIT (the
            # step no longer moves the point).  With fixLength these windows sit at their fixed point after four
            # to eight iterations; which evaluation is the first exact repeat there is the last bit's business --
            # the CPU oracle draws a third set of endings (scripts/dbg/lm4_fix_length.py).  Both must have arrived.
            assert opts.get("function_tolerance") == 0.0 and opts.get("fix_length")
            assert {(a.termination, a.reason), (b.termination, b.reason)} <= {(0, 4), (1, 1)}
            assert abs(a.final_cost - b.final_cost) <= 1e-13 * b.initial_cost
        # the same operations, but different summation orders (up to 256 windows the other path is the
        # four-wavefront kernel) and differently contracted multiply-adds; ten iterations amplify them
        assert abs(a.initial_cost - b.initial_cost) <= 1e-13 * b.initial_cost
        assert abs(a.final_cost - b.final_cost) <= 1e-9 * b.initial_cost
    # (window 30 of the batch has a flat direction: its parameters move by 1e-7 at a cost that agrees to 1e-14 between
    # the kernels and the oracle, scripts/dbg/lm4_vs_one_per_wave.py; the bar is the library's, 1e-6)
    assert np.abs(np.asarray(f4) - np.asarray(f1)).max() < 1e-6


def test_rows_that_stop_at_different_evaluations(solver, oracle):
    """Windows of one wavefront with very different histories: noise-free (stops at once on the gradient /
    function tolerance), noisy, and far-off starts -- every window against the oracle."""
    ws = []
    for i in range(12):
        ws.append(scene.make_window(8, 4, 90, seed=700 + i, noise_sigma=0.0 if i % 3 == 0 else 2e-3,
                                    init_angle_deg=3.0 if i % 4 else 20.0))
    opt = dict(max_num_iterations=30)
    fg, sg = _solve(solver, ws, assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_FOUR_PER_WAVE, **opt)
    its = set()
    for w, f, s in zip(ws, fg, sg):
        fo, so, _ = oracle.refine(w, oracle.make_options(**opt))
        assert (s.termination, s.reason) == (so.termination, so.reason)
        assert s.iterations == so.iterations
        assert abs(s.final_cost - so.final_cost) <= 1e-8 * max(so.initial_cost, 1e-300)
        assert np.abs(np.asarray(f)[: w.M, 9:13] - np.asarray(fo)[: w.M, 9:13]).max() < 1e-6
        its.add(s.iterations)
    assert len(its) >= 3  # the rows of a wavefront really did finish at different evaluations


def test_the_kernel_of_the_usual_window_equals_the_kernel_for_any_shape(solver):
    """k_lm4<5> (M = 5 refined frames, free lengths: compile-time constants) against k_lm4<0>, forced for the second
    handle by MVO_HIP_LM_GENERIC: same decisions, results equal to rounding."""
    import os
    os.environ["MVO_HIP_LM_GENERIC"] = "1"
    try:
        generic = capi.Solver(0)
    finally:
        del os.environ["MVO_HIP_LM_GENERIC"]
    try:
        for N, n in ((10, 9), (7, 4), (6, 6), (14, 5)):  # N - 1 = M: no long class at all (N = 6)
            ws = [scene.make_window(N, 5, 100, seed=900 + 7 * N + i) for i in range(n)]
            for opts in (FIXED10, dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                       parameter_tolerance=1e-8)):
                kw = dict(assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_FOUR_PER_WAVE, **opts)
                fa, sa = _solve(solver, ws, **kw)
                fb, sb = _solve(generic, ws, **kw)
                for a, b in zip(sa, sb):
                    assert (a.termination, a.reason, a.iterations, a.num_successful_steps) == \
                           (b.termination, b.reason, b.iterations, b.num_successful_steps), (N, opts)
                    assert abs(a.final_cost - b.final_cost) <= 1e-12 * b.initial_cost
                assert np.abs(np.asarray(fa) - np.asarray(fb)).max() < 1e-10
    finally:
        generic.close()


def test_two_windows_per_wavefront_equal_four(solver):
    """Batches of at most 2048 windows ride two per wavefront (rows 2 and 3 carry no window), larger ones four:
    MVO_HIP_LM4_WPW forces either.  A window's arithmetic does not depend on its row or on its neighbours: results
    identical, bit for bit, whatever the packing -- also for batch sizes that leave wavefronts half empty."""
    import os
    for n in (5, 9, 130):
        ws = _batch(n, seed0=1300)
        kw = dict(assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_FOUR_PER_WAVE, max_num_iterations=30)
        out = {}
        for wpw in ("2", "4"):
            os.environ["MVO_HIP_LM4_WPW"] = wpw
            try:
                out[wpw] = _solve(solver, ws, **kw)
            finally:
                del os.environ["MVO_HIP_LM4_WPW"]
        (f2, s2), (f4, s4) = out["2"], out["4"]
        assert np.array_equal(np.asarray(f2), np.asarray(f4))
        for a, b in zip(s2, s4):
            assert (a.termination, a.reason, a.iterations, a.final_cost) == (b.termination, b.reason, b.iterations, b.final_cost)

"""k_lm's compile-time specialisations (csrc/mvo_lm.h, "SPEC") against the generic kernel.

The host picks k_lm<NT, SPEC> from the solve's parameters: 1 / 2 no loss (resident / inline observations),
3 / 4 the same for windows with at most 32 long pairs per refined frame -- there k_lm<256> hands the numeric half
of the LM step to a class wavefront (LmSolver) and wavefront 0 keeps Ceres' control logic --, 5 the robust
path's relaunch.  A trace, fixLength or a partitioned solve fall back on SPEC 0, which is what most parity tests
(they ask for the trace) exercise.  Here the same windows are solved WITHOUT a trace by two handles of the same
library, one of them created under MVO_HIP_LM_GENERIC (SPEC 0 for everything): same verdicts, same
termination, results equal to rounding.  The LM loop restated: ceres::Solve, IterativeRefinement.cpp:231-233.
"""
import os

import numpy as np
import pytest

from _util import FIXED10, refined
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu


@pytest.fixture()
def generic_solver():
    os.environ["MVO_HIP_LM_GENERIC"] = "1"
    try:
        s = capi.Solver(0)
    finally:
        del os.environ["MVO_HIP_LM_GENERIC"]
    yield s
    s.close()


def same(a, sa, b, sb, M, tag, tol=1e-11):
    assert (sa.termination, sa.reason, sa.iterations) == (sb.termination, sb.reason, sb.iterations), tag
    assert (sa.num_successful_steps, sa.num_unsuccessful_steps) == (sb.num_successful_steps, sb.num_unsuccessful_steps), tag
    assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * max(sb.initial_cost, 1e-300), tag
    assert np.abs(refined(a, M) - refined(b, M)).max() <= tol, tag


SHAPES = [(3, 2, 60), (2, 1, 40), (6, 3, 80), (10, 5, 300), (12, 4, 100), (30, 5, 400),   # HALF layout: SPEC 3 / 4
          (40, 5, 200), (80, 5, 150), (120, 3, 120)]                                      # > 32 long pairs: SPEC 1 / 2


@pytest.mark.parametrize("mode", [capi.ASSEMBLY_AUTO, capi.ASSEMBLY_TILED])
def test_specialised_no_loss_solves_match_the_generic_kernel(solver, generic_solver, mode):
    """AUTO: the window's workgroup folds its own observations (SPEC 2 / 4); TILED: chip-wide fold (1 / 3)."""
    for i, (N, M, L) in enumerate(SHAPES):
        w = scene.make_window(N, M, L, seed=4100 + i)
        for kw in (FIXED10, dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                 parameter_tolerance=1e-8)):
            opt = capi.make_options(assembly_mode=mode, **kw)
            fa, sa = solver.refine(w, opt)
            fb, sb = generic_solver.refine(w, opt)
            same(fa, sa, fb, sb, M, (N, M, L, mode, kw["max_num_iterations"]))


def test_specialised_relaunch_of_the_robust_path_matches_the_generic_kernel(solver, generic_solver):
    """Cauchy loss and assembly_mode = PASSES on windows large enough for the chip-wide passes (SPEC 5)."""
    for i, (N, M, L) in enumerate([(30, 5, 1500), (60, 4, 900)]):
        w = scene.make_window(N, M, L, seed=4200 + i, outlier_fraction=0.2)
        for kw in (dict(use_loss_function=True), dict(assembly_mode=capi.ASSEMBLY_PASSES)):
            opt = capi.make_options(**kw, **FIXED10)
            fa, sa = solver.refine(w, opt)
            fb, sb = generic_solver.refine(w, opt)
            same(fa, sa, fb, sb, M, (N, M, kw))


def test_invalid_steps_take_the_same_road_in_both_kernels(solver, generic_solver):
    """A step whose model cost change is not positive is invalid (Ceres: HandleInvalidStep, five in a row end the
    solve as FAILURE).  The straight-line step publishes nothing then, the point in LDS is evaluated once more
    and the next verdict halves the radius (LmState::pending_invalid); in SPEC 3 / 4 the sign test runs on
    wavefront 0 behind barrier B2 while the solver wavefront's candidate is already being evaluated.  Rank-one
    windows whose cost underflows to 0 after ~20 accepted steps (tests/test_gpu_parity.py)."""
    from test_gpu_parity import _rank_one_window
    kw = dict(max_num_iterations=60, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    rng = np.random.default_rng(11)
    failed = 0
    for i in range(12):
        v0 = [rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6), 0.8]
        N, M = [(2, 1), (3, 2), (8, 3)][i % 3]
        w = _rank_one_window(v0, N, M, n=[8, 1, 3, 50][i % 4])
        for mode in (capi.ASSEMBLY_AUTO, capi.ASSEMBLY_TILED):
            opt = capi.make_options(assembly_mode=mode, **kw)
            fa, sa = solver.refine(w, opt)
            fb, sb = generic_solver.refine(w, opt)
            assert (sa.termination, sa.reason) == (sb.termination, sb.reason), (i, mode)
            if sa.termination == 2:
                failed += 1
                assert sa.reason == 6 and sa.iterations == sb.iterations, (i, mode)
                assert np.array_equal(fa[:M, 9:12], w.frames[:M, 9:12])
    assert failed >= 8

"""The robust path of one window as ONE persistent launch (k_lm<256, 6>, csrc/mvo_lm.h, csrc/mvo_grid.h) against
the launch-per-evaluation path of the same library.

With a loss function (or assembly_mode = PASSES) every evaluation is a pass over the residual blocks
(IterativeRefinement.cpp:85-88 + :231-233).  When the pass grid of a single window fits the chip in one round the
library keeps that grid for the whole solve: every workgroup runs the LM step itself (replicated, bit for bit the
same everywhere), passes over its own tiles, and one bounded grid-wide barrier per evaluation hands the pairs'
projections round.  The same arithmetic in the same order as k_pass + k_lm<256, 5> launch by launch -- a handle
created under MVO_HIP_NO_PERSISTENT -- so verdicts, termination and costs must agree and the refined parameters
to rounding; both are held against the oracle too.  A launch that ends without a result (a wait gave up) makes the
host repeat the solve launch by launch: MVO_HIP_PERSISTENT_TEST_FAIL provokes exactly that.
"""
import os

import numpy as np
import pytest

from _util import FIXED10, refined
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu


def _handle(var):
    os.environ[var] = "1"
    try:
        return capi.Solver(0)
    finally:
        del os.environ[var]


@pytest.fixture()
def classic_solver():
    s = _handle("MVO_HIP_NO_PERSISTENT")
    yield s
    s.close()


CASES = [(30, 5, 1500, 4300), (30, 5, 5000, 1005), (12, 3, 2500, 4301), (60, 4, 900, 4302), (100, 5, 400, 4303), (8, 2, 3000, 4304)]


@pytest.mark.parametrize("kw", [dict(use_loss_function=True), dict(assembly_mode=capi.ASSEMBLY_PASSES)])
def test_persistent_robust_solve_matches_the_launch_per_evaluation_path(solver, classic_solver, oracle, kw):
    for N, M, L, seed in CASES:
        w = scene.make_window(N, M, L, seed=seed, outlier_fraction=0.2, span=("sliding", 20) if N >= 30 else "full")
        for tol in (FIXED10, dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8)):
            opt = capi.make_options(**kw, **tol)
            fa, sa = solver.refine(w, opt)
            fb, sb = classic_solver.refine(w, opt)
            tag = (N, M, L, kw, tol["max_num_iterations"])
            assert (sa.termination, sa.reason, sa.iterations) == (sb.termination, sb.reason, sb.iterations), tag
            assert (sa.num_successful_steps, sa.num_unsuccessful_steps) == (sb.num_successful_steps, sb.num_unsuccessful_steps), tag
            assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * sb.initial_cost, tag
            assert np.abs(refined(fa, M) - refined(fb, M)).max() <= 1e-11, tag
        fo, so, _ = oracle.refine(w, oracle.make_options(**{k: v for k, v in kw.items() if k == "use_loss_function"}, **FIXED10))
        fa, sa = solver.refine(w, capi.make_options(**kw, **FIXED10))
        assert np.abs(refined(fa, M) - refined(fo, M)).max() < 1e-6, (N, M, L)
        assert abs(sa.final_cost - so.final_cost) <= 1e-9 * so.initial_cost


def test_repeated_persistent_solves_on_one_handle(solver, classic_solver):
    """The barrier words are cleared per launch, tickets and slots are left clean: solve after solve, and a
    launch-per-evaluation solve (a traced one) in between, give the same answer."""
    w = scene.make_window(30, 5, 2000, seed=4310, outlier_fraction=0.2, span=("sliding", 20))
    opt = capi.make_options(use_loss_function=True, **FIXED10)
    ref, sr = classic_solver.refine(w, opt)
    for i in range(6):
        if i == 3:
            solver.set_trace(32)  # a trace is not the persistent kernel's business: SPEC 0, launch by launch
            f, s_ = solver.refine(w, opt)
            assert len(solver.trace(32)) == s_.iterations + 1
            solver.set_trace(0)
        else:
            f, s_ = solver.refine(w, opt)
        assert s_.iterations == sr.iterations and np.abs(refined(f, 5) - refined(ref, 5)).max() <= 1e-11, i


def test_a_persistent_launch_without_a_result_is_repeated_launch_by_launch(classic_solver):
    s = _handle("MVO_HIP_PERSISTENT_TEST_FAIL")
    try:
        w = scene.make_window(30, 5, 1500, seed=4320, outlier_fraction=0.2, span=("sliding", 20))
        opt = capi.make_options(use_loss_function=True, **FIXED10)
        fa, sa = s.refine(w, opt)
        fb, sb = classic_solver.refine(w, opt)
        assert sa.iterations == sb.iterations == 10
        assert np.array_equal(refined(fa, 5), refined(fb, 5))
        fa2, _ = s.refine(w, opt)  # the handle has stopped trying: straight to the launch-per-evaluation path
        assert np.array_equal(refined(fa2, 5), refined(fb, 5))
    finally:
        s.close()

"""Fold and LM loop in one launch (k_lm<256, 7>, csrc/mvo_lm.h) against the two launches they replace.

One small window without a loss whose fold takes at most one workgroup per CU: the grid is the fold's plus one, the
fold workgroups fold their tiles and leave through the ticket tail as in k_fold, workgroup 0 sets the LM loop up, waits
(bounded) until every pair is finished and runs the LM loop of k_lm<256, 3>.  Same sums, added up in the same order, same LM code: the results must
be IDENTICAL to those of a handle created under MVO_HIP_NO_FUSED.  A wait that gives up ends without a result and the
host repeats the solve as two launches (MVO_HIP_FUSED_TEST_FAIL).  Replaces ceres::Solve,
IterativeRefinement.cpp:231-233, for the reference's single-window call (Refiner.cpp:85-92)."""
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


def _same(fa, sa, fb, sb, M):
    assert (sa.termination, sa.reason, sa.iterations, sa.num_successful_steps, sa.num_unsuccessful_steps) == \
           (sb.termination, sb.reason, sb.iterations, sb.num_successful_steps, sb.num_unsuccessful_steps)
    assert sa.initial_cost == sb.initial_cost and sa.final_cost == sb.final_cost
    assert np.array_equal(refined(fa, M), refined(fb, M))


SHAPES = [(30, 5, 5000), (10, 5, 300), (12, 4, 2000), (30, 5, 700), (20, 3, 1500)]


def test_one_launch_equals_two_launches(solver):
    apart = _handle("MVO_HIP_NO_FUSED")
    try:
        for i, (N, M, L) in enumerate(SHAPES):
            w = scene.make_window(N, M, L, seed=5100 + i)
            for kw in (FIXED10, dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                     parameter_tolerance=1e-8)):
                opt = capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **kw)
                fa, sa = solver.refine(w, opt)
                fb, sb = apart.refine(w, opt)
                _same(fa, sa, fb, sb, M)
    finally:
        apart.close()


def test_split_phase_counts_one_launch_and_repeats(solver):
    """upload once, solve many times: the count of finished pairs is never reset on the device (every launch waits for
    its own target); an event-timed solve in between takes the two launches (the fold's duration is wanted there)."""
    w = scene.config2()
    solver.upload([w], capi.make_options(**FIXED10))
    solver.set_profiling(False)
    solver.solve()
    f0, s0 = solver.download()
    assert solver.timing()["lm_launches"] == 1 and solver.timing()["assembly_launches"] == 0
    for i in range(20):
        if i == 7:
            solver.set_profiling(True)
        solver.solve()
        if i == 7:
            assert solver.timing()["assembly_launches"] == 1
            solver.set_profiling(False)
        f, s = solver.download()
        _same(f, s[0], f0, s0[0], w.M)


def test_against_the_oracle(solver, oracle):
    w = scene.make_window(16, 5, 1200, seed=5200)
    opt = dict(max_num_iterations=30)
    f, s = solver.refine(w, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **opt))
    fo, so, _ = oracle.refine(w, oracle.make_options(**opt))
    assert (s.termination, s.reason, s.iterations) == (so.termination, so.reason, so.iterations)
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.initial_cost
    assert np.abs(refined(f, w.M) - refined(fo, w.M)).max() < 1e-6


def test_a_wait_that_gives_up_falls_back_on_two_launches(solver):
    failing = _handle("MVO_HIP_FUSED_TEST_FAIL")
    try:
        w = scene.make_window(30, 5, 3000, seed=5300)
        opt = capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **FIXED10)
        fa, sa = failing.refine(w, opt)     # workgroup 0 gives up, the host repeats the solve
        fb, sb = solver.refine(w, opt)
        _same(fa, sa, fb, sb, w.M)
        fa, sa = failing.refine(w, opt)     # ... and stays on two launches
        _same(fa, sa, fb, sb, w.M)
    finally:
        failing.close()


def test_a_window_of_a_few_thousand_blocks_takes_the_fused_launch_by_default(oracle):
    """Between 3072 and 8192 residual blocks a single window without a loss is folded chip-wide inside the fused launch
    (no event records at upload time) instead of by its own workgroup: one launch, no assembly launch, same results as
    the inline fold of a handle that records events (to rounding: the sums are cut differently) and as the oracle."""
    w = scene.make_window(10, 5, 150, seed=5400)
    assert 3072 < w.n_residual_blocks <= 8192
    opt = capi.make_options(max_num_iterations=30)
    with capi.Solver(0, profiling=False) as quiet, capi.Solver(0, profiling=True) as timed:
        fa, sa = quiet.refine(w, opt)
        assert quiet.timing()["lm_launches"] == 1 and quiet.timing()["assembly_launches"] == 0
        fb, sb = timed.refine(w, opt)
    fo, so, _ = oracle.refine(w, oracle.make_options(max_num_iterations=30))
    for f, s in ((fa, sa), (fb, sb)):
        assert (s.termination, s.reason, s.iterations) == (so.termination, so.reason, so.iterations)
        assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.initial_cost
        assert np.abs(refined(f, w.M) - refined(fo, w.M)).max() < 1e-6

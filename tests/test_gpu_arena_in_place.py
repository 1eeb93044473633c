"""One small window without a loss (the reference's shipped call: numberToNote = 3, numberToRefine = 2, <= 60 features,
Refiner.cpp:85-92 -> IterativeRefinement.cpp:61-253): the kernel reads frames, starting point and rays where the host
wrote them -- the handle's pinned arena, mapped into the device's address space -- instead of behind a copy command
(mvo_hip_upload, `arena_in_place`).  Same kernel, same data: the results must be IDENTICAL to those of a handle created
under MVO_HIP_ARENA_COPY, call after call with CHANGING data at the same host addresses (nothing stale may be read)."""
import os

import numpy as np
import pytest

from _util import refined
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu

SHIPPED = dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8)


def _copy_handle():
    os.environ["MVO_HIP_ARENA_COPY"] = "1"
    try:
        return capi.Solver(0)
    finally:
        del os.environ["MVO_HIP_ARENA_COPY"]


def _same(fa, sa, fb, sb, M):
    assert (sa.termination, sa.reason, sa.iterations, sa.num_successful_steps, sa.num_unsuccessful_steps) == \
           (sb.termination, sb.reason, sb.iterations, sb.num_successful_steps, sb.num_unsuccessful_steps)
    assert sa.initial_cost == sb.initial_cost and sa.final_cost == sb.final_cost
    assert np.array_equal(refined(fa, M), refined(fb, M))


def test_changing_windows_at_the_same_addresses(solver):
    """sixty different windows of the shipped shape, one after the other through the same handle (same arena, same
    addresses), twice round: every call against the copying handle, bit for bit."""
    copy = _copy_handle()
    try:
        ws = [scene.make_window(3, 2, 20 + (7 * i) % 41, seed=8100 + i) for i in range(60)]
        opt = capi.make_options(**SHIPPED)
        want = [copy.refine(w, opt) for w in ws]
        for _ in range(2):
            for w, (fb, sb) in zip(ws, want):
                fa, sa = solver.refine(w, opt)
                _same(fa, sa, fb, sb, w.M)
    finally:
        copy.close()


@pytest.mark.parametrize("shape", [(3, 2), (4, 2), (4, 1), (3, 1), (5, 1)])
def test_shapes_with_at_most_eight_pairs(solver, oracle, shape):
    """every shape the in-place path takes (at most eight frame pairs), with fix_length and without, against the copying
    handle and the oracle"""
    N, M = shape
    assert scene.num_pairs(M, N) <= 8
    copy = _copy_handle()
    try:
        for fix in (False, True):
            w = scene.make_window(N, M, 60, seed=8200 + 10 * N + M)
            kw = dict(fix_length=fix, **SHIPPED)
            fa, sa = solver.refine(w, capi.make_options(**kw))
            fb, sb = copy.refine(w, capi.make_options(**kw))
            _same(fa, sa, fb, sb, M)
            fo, so, _ = oracle.refine(w, oracle.make_options(**kw))
            assert (sa.termination, sa.reason, sa.iterations) == (so.termination, so.reason, so.iterations)
            assert abs(sa.final_cost - so.final_cost) <= 1e-9 * max(so.initial_cost, 1e-300)
            assert np.abs(refined(fa, M) - refined(fo, M)).max() < 1e-6
    finally:
        copy.close()


def test_split_phase_solves_again_from_the_arena(solver):
    """upload once, solve many times: every solve reads the arena again; a window with a loss (copied) in between"""
    w = scene.reference_defaults_window()
    solver.upload([w], capi.make_options(**SHIPPED))
    solver.solve()
    f0, s0 = solver.download()
    f0 = np.array(f0[0])
    for _ in range(10):
        solver.solve()
        f, s = solver.download()
        _same(f[0], s[0], f0, s0[0], w.M)
    solver.refine(w, capi.make_options(use_loss_function=True, **SHIPPED))
    f1, s1 = solver.refine(w, capi.make_options(**SHIPPED))
    _same(f1, s1, f0, s0[0], w.M)
    assert np.abs(refined(f0, w.M)).max() > 0.0


@pytest.mark.parametrize("loss", [False, True])
def test_rays_of_a_mid_size_window_read_in_the_arena_by_the_fold(solver, loss):
    """One-shot call, one window folded chip-wide with at most 65536 residual blocks: the fold reads the gathered rays in
    the pinned arena (every ray exactly once) and only the tables are copied.  Different windows one after the other
    through the same handle, bit for bit against a copying handle."""
    copy = _copy_handle()
    try:
        ws = [scene.make_window(10, 5, 300 + 40 * (i % 5), seed=8300 + i) for i in range(12)]
        opt = capi.make_options(use_loss_function=loss, **SHIPPED)
        want = [copy.refine(w, opt) for w in ws]
        for _ in range(2):
            for w, (fb, sb) in zip(ws, want):
                fa, sa = solver.refine(w, opt)
                _same(fa, sa, fb, sb, w.M)
    finally:
        copy.close()

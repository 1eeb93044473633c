"""Batches of windows with at most 64 frame pairs: the LM kernels (k_lm<256>, k_lm<64>, k_lm4) add the fold's slots up
themselves and the fold runs without its ticket tails, cut finer (mvo_hip_upload, "slots plan"), against the plan it
replaces -- fold with tails, per-pair sums, forced for a second handle by MVO_HIP_LM4_SUMS -- and against the oracle.
Same observations, same LM code; only the partial sums are cut differently, so the decisions must be the same and the
results equal to rounding.  Replaces ceres::Solve over the residual blocks of IterativeRefinement.cpp:113-229 for
batched windows (mvo_hip_refine_batch)."""
import os

import numpy as np
import pytest

from _util import FIXED10, refined
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu


def _sums_handle():
    os.environ["MVO_HIP_LM4_SUMS"] = "1"
    try:
        return capi.Solver(0)
    finally:
        del os.environ["MVO_HIP_LM4_SUMS"]


def _close(sa, sb, fa, fb, M):
    for a, b in zip(sa, sb):
        assert (a.termination, a.reason, a.iterations, a.num_successful_steps, a.num_unsuccessful_steps) == \
               (b.termination, b.reason, b.iterations, b.num_successful_steps, b.num_unsuccessful_steps)
        assert abs(a.initial_cost - b.initial_cost) <= 1e-13 * b.initial_cost
        assert abs(a.final_cost - b.final_cost) <= 1e-10 * b.initial_cost
    for x, y in zip(fa, fb):
        assert np.abs(refined(x, M) - refined(y, M)).max() < 1e-6


# 2..256 windows: k_lm<256>, one workgroup per window; 257..640: k_lm<64>; from 641: k_lm4
@pytest.mark.parametrize("n", [2, 9, 200, 300, 700])
def test_slots_read_by_the_lm_kernels_equal_the_pair_sums_of_the_tails(solver, n):
    base = [scene.config3_window(i) for i in range(8)]
    ws = [base[i % 8] for i in range(n)]
    sums = _sums_handle()
    try:
        for kw in (FIXED10, dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                 parameter_tolerance=1e-8)):
            opt = capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **kw)
            fa, sa = solver.refine_batch(ws, opt)
            assert solver.timing()["assembly_launches"] == 1 and solver.timing()["lm_launches"] == 1
            fb, sb = sums.refine_batch(ws, opt)
            _close(sa, sb, fa, fb, ws[0].M)
    finally:
        sums.close()


def test_windows_of_very_different_lengths(solver, oracle):
    """One window a hundred times longer than the others: the cut follows the longest pair (a pair's slots are added up
    by one lane), or the upload keeps the tails -- either way every window against the oracle."""
    ws = [scene.make_window(8, 4, 60, seed=7600 + i) for i in range(11)]
    ws.insert(5, scene.make_window(8, 4, 6000, seed=7650))
    opt = dict(max_num_iterations=30)
    fg, sg = solver.refine_batch(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **opt))
    for w, f, s in zip(ws, fg, sg):
        fo, so, _ = oracle.refine(w, oracle.make_options(**opt))
        assert (s.termination, s.reason, s.iterations) == (so.termination, so.reason, so.iterations)
        assert abs(s.final_cost - so.final_cost) <= 1e-9 * max(so.initial_cost, 1e-300)
        assert np.abs(refined(f, w.M) - refined(fo, w.M)).max() < 1e-6


@pytest.mark.parametrize("mode", [0, capi.ASSEMBLY_ONE_PER_WAVE, capi.ASSEMBLY_FOUR_PER_WAVE])
def test_pairs_without_observations_in_the_batch(solver, mode):
    """Windows whose far pairs have no observations at all (no slot) beside ordinary ones, under each of the three LM
    kernels: an empty pair's sums are zero, whichever plan; it must not read a slot of its neighbours."""
    ws = [scene.make_window(10, 5, 150, seed=7700 + i) for i in range(6)]
    thin = [scene.make_window(10, 5, 200, seed=7750 + i, span=("anchored", 6, 8)) for i in range(6)]
    assert all(np.any(np.diff(w.pair_offsets) == 0) for w in thin)
    batch = [w for pair in zip(ws, thin) for w in pair]
    sums = _sums_handle()
    try:
        opt = capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | mode, **FIXED10)
        fa, sa = solver.refine_batch(batch, opt)
        fb, sb = sums.refine_batch(batch, opt)
        _close(sa, sb, fa, fb, 5)
    finally:
        sums.close()


@pytest.mark.parametrize("n", [65, 300, 640, 1000])
def test_records_written_into_the_host_mirror_equal_copied_records(solver, n):
    """Up to 1024 windows the LM kernels write the result records and their marks into the pinned host mirror and
    the host polls the marks (64 windows until the end of round 3); a handle created under MVO_HIP_NO_HOST_RESULTS has
    them copied by the stream.  Same kernels, same data: identical results.  (mvo_hip_refine_batch; the write-back of
    IterativeRefinement.cpp:236-252 per window.)"""
    base = [scene.config3_window(i) for i in range(8)]
    ws = [base[i % 8] for i in range(n)]
    os.environ["MVO_HIP_NO_HOST_RESULTS"] = "1"
    try:
        copied = capi.Solver(0)
    finally:
        del os.environ["MVO_HIP_NO_HOST_RESULTS"]
    try:
        opt = capi.make_options(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                parameter_tolerance=1e-8)
        fa, sa = solver.refine_batch(ws, opt)
        fb, sb = copied.refine_batch(ws, opt)
        assert np.array_equal(np.asarray(fa), np.asarray(fb))
        for a, b in zip(sa, sb):
            assert (a.termination, a.reason, a.iterations, a.initial_cost, a.final_cost) == \
                   (b.termination, b.reason, b.iterations, b.initial_cost, b.final_cost)
    finally:
        copied.close()

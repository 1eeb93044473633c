"""The observation passes' ticket tails (csrc/mvo_kernels.h: block_store_partials / block_finish_pairs)
against k_reduce_slots, BIT FOR BIT.

The workgroup that stores the last slot of a frame pair adds that pair's slots up.  Slots are stored by
threads item = tile * 8 + value, so from nine tiles per workgroup on wavefronts 1..3 store slots while
wavefront 0 takes the tickets: every wavefront has to wait for its own stores (s_waitcnt vmcnt(0)) before
the barrier that precedes the tickets.  These windows are shaped so that a workgroup holds many short
tiles (pairs far shorter than a wavefront's chunk) AND some pairs span several workgroups (several slots,
added up by whichever workgroup comes last).  MVO_HIP_NO_TAILS makes the same library add the slots with
a k_reduce_slots launch behind every pass instead: same slots, same order, so sums and results must be
identical to the last bit.  Runs on poisoned LDS and 0xFF-poisoned allocations like every GPU test.
Accumulation in the reference: Ceres' AddResidualBlock loop, IterativeRefinement.cpp:369-455.
"""
import os

import numpy as np
import pytest

from _util import FIXED10
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu


def ragged_windows():
    # anchored spans: pair (a, b) holds the landmarks that reach back to frame b + 1 -- hundreds of
    # residual blocks for small b, a handful for b near N - 2
    return [scene.make_window(100, 5, 700, seed=300 + i, span=("anchored", 2, 100), outlier_fraction=0.1)
            for i in range(3)]


@pytest.fixture()
def reference_solver():
    """A second handle of the same library whose passes have no tails (k_reduce_slots launches)."""
    os.environ["MVO_HIP_NO_TAILS"] = "1"
    try:
        s = capi.Solver(0)
    finally:
        del os.environ["MVO_HIP_NO_TAILS"]
    yield s
    s.close()


def tile_census(w):
    """(largest number of tiles in one workgroup, largest number of slots of one pair) as do_upload cuts
    one window of this shape: chunks of >= 256 residual blocks per wavefront, 4 wavefronts per workgroup."""
    counts = np.diff(w.pair_offsets.astype(np.int64))
    n = int(counts.sum())
    waves = min(max((n + 255) // 256, 1), 256 * 16)
    chunk = max((n + waves - 1) // waves, 1)
    tiles_of_wave, fill, wave = [0], 0, 0
    slots = []
    for c in counts:
        done, first_block = 0, None
        blocks = set()
        while done < c:
            if fill >= chunk:
                tiles_of_wave.append(0)
                fill = 0
            take = min(c - done, chunk - fill)
            tiles_of_wave[-1] += 1
            blocks.add((len(tiles_of_wave) - 1) // 4)
            done += take
            fill += take
        slots.append(len(blocks))
    per_block = [sum(tiles_of_wave[i:i + 4]) for i in range(0, len(tiles_of_wave), 4)]
    return max(per_block), max(slots)


def test_the_windows_exercise_what_they_should():
    w = ragged_windows()[0]
    tiles, slots = tile_census(w)
    assert tiles >= 9    # wavefronts 1..3 store slots
    assert slots >= 2    # some pair is added up across workgroups


@pytest.mark.parametrize("use_loss", [False, True])
def test_tails_equal_reduce_slots_bit_for_bit(solver, reference_solver, use_loss):
    ws = ragged_windows()
    opt = capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, use_loss_function=use_loss, **FIXED10)
    out = []
    for s in (solver, reference_solver):
        s.upload(ws, opt)
        for _ in range(3):  # the tickets must be back at zero after every launch
            s.solve()
        frames, summ = s.download()
        out.append((s.pair_sums().copy(), frames.copy(), [(x.termination, x.iterations, x.final_cost) for x in summ]))
    (sums_t, frames_t, summ_t), (sums_r, frames_r, summ_r) = out
    assert np.isfinite(sums_t).all()
    assert np.array_equal(sums_t, sums_r)
    assert np.array_equal(frames_t, frames_r)
    assert summ_t == summ_r


def test_fold_tail_equals_slot_sums_of_the_plain_fold(solver):
    """No loss: the fused fold's per-pair sums against the un-fused fold's slots added up on the host in
    slot order (mvo_hip_pair_moments), for every pair of every window."""
    ws = ragged_windows()
    solver.upload(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **FIXED10))
    solver.solve()
    sums = solver.pair_sums()
    assert np.array_equal(sums[:, :6], solver.pair_moments())

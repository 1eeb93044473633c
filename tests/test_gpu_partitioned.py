"""Observation-partitioned solve (one large window over several GPUs) on ONE
GPU: two handles play two ranks (threads + a barrier stand in for the all-reduce), and a
world-size-1 RCCL process group exercises the real torch.distributed call path."""
import threading

import numpy as np
import pytest
import torch

from _util import FIXED10, refined
from mvo_amd import capi, scene, sharding

pytestmark = pytest.mark.gpu


def _two_rank_solve(w, opt):
    world = 2
    solvers = [capi.Solver(0) for _ in range(world)]
    sums = [None] * world  # what the ranks exchange: 8 doubles per pair, or the contracted class totals with a loss
    barrier = threading.Barrier(world)
    errors = []

    def allreduce_for(rank):
        def allreduce(t):
            torch.cuda.synchronize()
            barrier.wait()
            total = sums[0] + sums[1]
            torch.cuda.synchronize()
            barrier.wait()
            t.copy_(total)
            torch.cuda.synchronize()
            barrier.wait()
        return allreduce

    def run(rank):
        try:
            solvers[rank].upload([sharding.shard_observations(w, rank, world)], opt)
            n = solvers[rank].partial_sums_count()
            assert n == (capi.CLASS_DOUBLES if opt.use_loss_function else 8 * w.n_pairs)
            sums[rank] = torch.zeros(n, dtype=torch.float64, device="cuda")
            solvers[rank].solve_partitioned(allreduce_for(rank), sums[rank])
        except Exception as e:  # pragma: no cover
            errors.append(e)
            barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    out = [s.download() for s in solvers]
    for s in solvers:
        s.close()
    return out


@pytest.mark.parametrize("use_loss", [False, True])
def test_two_shards_equal_single_solve(solver, oracle, use_loss):
    w = scene.make_window(30, 5, 800, seed=41, span=("sliding", 18))
    kw = dict(use_loss_function=use_loss, **FIXED10)
    out = _two_rank_solve(w, capi.make_options(**kw))
    (f0, s0), (f1, s1) = out
    assert np.array_equal(f0, f1) and s0[0].final_cost == s1[0].final_cost   # ranks agree bitwise
    fs, ss = solver.refine(w, capi.make_options(**kw))
    assert np.abs(refined(f0[0], 5) - refined(fs, 5)).max() < 1e-9           # summation order only
    fo, so, _ = oracle.refine(w, oracle.make_options(**kw))
    assert np.abs(refined(f0[0], 5) - refined(fo, 5)).max() < 1e-6
    assert so.iterations == 10
    if s0[0].iterations != 10:
        # tolerances 0: the loop can only stop early on an exactly zero cost change or step,
        # i.e. once the iterate is converged to the last bit (rounding decides when that happens)
        assert s0[0].termination == 0 and s0[0].iterations >= 5
        assert abs(s0[0].final_cost - so.final_cost) <= 1e-12 * so.final_cost


@pytest.mark.parametrize("use_loss", [True, False])
def test_contracted_class_totals_match_numpy(solver, use_loss):
    """What a rank hands to the all-reduce in pass mode (k_fold tail -> project_pair8 -> k_class_sums) against
    the numpy statement of the same sums at the starting point, for a window whose long classes span more
    than one lane round (N - 1 - M > 64) and for a batch of short ones."""
    mode = 0 if use_loss else capi.ASSEMBLY_PASSES
    for ws in ([scene.make_window(80, 4, 300, seed=45, span=("anchored", 2, 80), outlier_fraction=0.1)],
               [scene.config3_window(i) for i in range(5)]):
        solver.upload(ws, capi.make_options(use_loss_function=use_loss, assembly_mode=mode, **FIXED10))
        n = solver.partial_sums_count()
        assert n == capi.CLASS_DOUBLES * len(ws)
        buf = torch.zeros(n, dtype=torch.float64, device="cuda")
        assert solver._lib.mvo_hip_partial_begin(solver._h) == capi.OK
        assert solver._lib.mvo_hip_partial_sums(solver._h, buf.data_ptr(), n, 0) == capi.OK
        got = buf.cpu().numpy().reshape(len(ws), capi.CLASS_DOUBLES)
        for i, w in enumerate(ws):
            ref = sharding.contracted_class_totals(w, use_loss)
            assert np.abs(got[i] - ref).max() <= 1e-11 * np.abs(ref).max()


def test_default_tolerances_terminate_on_all_ranks(oracle):
    w = scene.make_window(9, 3, 300, seed=42)
    out = _two_rank_solve(w, capi.make_options(use_loss_function=True))
    fo, so, _ = oracle.refine(w, oracle.make_options(use_loss_function=True))
    for f, s in out:
        assert s[0].termination == so.termination == 0
        assert abs(s[0].iterations - so.iterations) <= 1
        assert np.abs(refined(f[0], 3)[:, :3] - refined(fo, 3)[:, :3]).max() < 1e-4


def test_rccl_world1_call_path(solver):
    """torch.distributed backend "nccl" is RCCL on ROCm; one rank is all a one-GPU box offers."""
    import os
    import socket
    import torch.distributed as dist

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        w = scene.config1()
        for use_loss in (False, True):
            opt = capi.make_options(use_loss_function=use_loss, **FIXED10)
            def allreduce(t):
                dist.all_reduce(t)
                torch.cuda.synchronize()

            solver.upload([w], opt)
            sums = torch.zeros(solver.partial_sums_count(), dtype=torch.float64, device="cuda")
            solver.solve_partitioned(allreduce, sums)
            fp, sp = solver.download()
            fs, ss = solver.refine(w, opt)
            assert np.abs(refined(fp[0], 5) - refined(fs, 5)).max() < 1e-9
            assert float(sums.abs().sum()) > 0
    finally:
        dist.destroy_process_group()

"""Landmark-partitioned bundle adjustment (include/mvo_hip_ba.h, mvo_hip_ba_partial_*) on ONE GPU:
two handles play two ranks (threads + a barrier stand in for the all-reduce of the reduced camera
system), and a world-size-1 RCCL group exercises the real torch.distributed call path.
PARITY UNPINNED (no reference counterpart): checked against the single-handle solve and the
self-oracle."""
import threading

import numpy as np
import pytest
import torch

import mvo_ba_oracle_py as bo
from mvo_amd import ba_capi, ba_scene, sharding

pytestmark = pytest.mark.gpu
FIXED = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


def _two_rank_solve(sc, opt):
    world = 2
    solvers = [ba_capi.BaSolver(0) for _ in range(world)]
    n0 = 18 * sc.M * (sc.M + 1) + 18 * sc.M + 1
    bufs = [torch.zeros(n0, dtype=torch.float64, device="cuda") for _ in range(world)]
    barrier = threading.Barrier(world)
    errors = []

    def reducer(rank, op):
        def allreduce(t):  # t is a view of bufs[rank]
            n = t.numel()
            torch.cuda.synchronize()
            barrier.wait()
            total = op(bufs[0][:n], bufs[1][:n])
            torch.cuda.synchronize()
            barrier.wait()
            t.copy_(total)
            torch.cuda.synchronize()
            barrier.wait()
        return allreduce

    shards = [sharding.shard_landmarks(sc, r, world) for r in range(world)]

    def run(rank):
        try:
            solvers[rank].upload(shards[rank][0], opt)
            solvers[rank].solve_partitioned(reducer(rank, torch.add), reducer(rank, torch.maximum), bufs[rank])
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
    return out, shards


@pytest.mark.parametrize("loss", [0, 1])
def test_two_landmark_shards_equal_single_solve(loss):
    sc = ba_scene.make_ba_scene(20, 10, 400, seed=31, span=9)
    kw = dict(max_num_iterations=6, loss=loss, loss_scale=0.003, **FIXED)
    out, shards = _two_rank_solve(sc, ba_capi.make_options(**kw))
    (p0, X0, s0), (p1, X1, s1) = out
    assert np.array_equal(p0, p1) and s0.final_cost == s1.final_cost and s0.iterations == s1.iterations == 6
    with ba_capi.BaSolver(0) as single:
        ps, Xs, ss = single.refine(sc, ba_capi.make_options(**kw))
    X = np.concatenate([X0, X1])
    assert len(X) == sc.n_landmarks
    assert np.abs(p0 - ps).max() < 1e-9 and np.abs(X - Xs).max() < 1e-8      # summation order only
    assert abs(s0.final_cost - ss.final_cost) <= 1e-9 * ss.final_cost
    po, Xo, so, _ = bo.refine(sc, bo.make_options(**kw))
    assert np.abs(p0 - po).max() < 1e-6 and np.abs(X - Xo).max() < 1e-6 * np.abs(Xo).max()
    assert abs(s0.initial_cost - so.initial_cost) <= 1e-11 * so.initial_cost


def test_default_tolerances_terminate_on_all_ranks():
    sc = ba_scene.make_ba_scene(12, 8, 300, seed=21, span=8)
    out, _ = _two_rank_solve(sc, ba_capi.make_options(max_num_iterations=50))
    _, _, so, _ = bo.refine(sc, bo.make_options(max_num_iterations=50))
    for _, _, s in out:
        assert s.termination == so.termination == 0 and s.iterations == so.iterations


def test_rccl_world1_call_path():
    import os
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sc = ba_scene.ba_config1()
        kw = dict(max_num_iterations=5, **FIXED)
        buf = torch.zeros(18 * sc.M * (sc.M + 1) + 18 * sc.M + 1, dtype=torch.float64, device="cuda")

        def ar(op):
            def f(t):
                dist.all_reduce(t, op=op)
                torch.cuda.synchronize()
            return f

        with ba_capi.BaSolver(0) as s:
            s.upload(sc, ba_capi.make_options(**kw))
            s.solve_partitioned(ar(dist.ReduceOp.SUM), ar(dist.ReduceOp.MAX), buf)
            p, X, sm = s.download()
            s.solve()
            p2, X2, sm2 = s.download()
        assert np.array_equal(p, p2) and np.array_equal(X, X2) and sm.final_cost == sm2.final_cost
    finally:
        if created:
            dist.destroy_process_group()

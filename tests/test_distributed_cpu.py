"""world_size-2 gloo tests (CPU) of the multi-GPU partition plans in mvo_amd/sharding.py.

The HIP kernels cannot run here, so the per-rank compute is the CPU oracle; what is under test is
the host logic of the N>1 path: the shards tile the work exactly, per-pair sums are additive
under all-reduce, and every rank ends with the same numbers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvo_amd import scene, sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import mvo_oracle_py as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) one large window split by observation range: all-reduce of the per-pair moments
        w = scene.make_window(12, 4, 300, seed=77, span=("anchored", 2, 12))
        shard = sharding.shard_observations(w, rank, world)
        local = torch.from_numpy(orc.pair_moments(shard).copy())
        n_local = torch.tensor([shard.n_residual_blocks], dtype=torch.int64)
        dist.all_reduce(local)
        dist.all_reduce(n_local)
        full = orc.pair_moments(w)
        ok_moments = bool(np.abs(local.numpy() - full).max() <= 1e-12 * np.abs(full).max())
        ok_count = int(n_local.item()) == w.n_residual_blocks
        # (1b) the same window under the Cauchy loss: the ranks exchange the CONTRACTED sums (class totals of
        # the pairs' projected moments, 352 doubles) -- additive because the projection is linear in S
        wl = scene.make_window(12, 4, 300, seed=78, span=("anchored", 2, 12), outlier_fraction=0.2)
        mine_cls = torch.from_numpy(sharding.contracted_class_totals(sharding.shard_observations(wl, rank, world), True))
        dist.all_reduce(mine_cls)
        full_cls = sharding.contracted_class_totals(wl, True)
        ok_moments = ok_moments and mine_cls.numel() == sharding.CLASS_DOUBLES and bool(
            np.abs(mine_cls.numpy() - full_cls).max() <= 1e-12 * np.abs(full_cls).max())
        # (2) independent windows: block partition, results gathered
        ids = list(sharding.shard_windows(7, rank, world))
        opt = orc.make_options(max_num_iterations=5, function_tolerance=0.0, gradient_tolerance=0.0,
                               parameter_tolerance=0.0)
        mine = torch.zeros(7, 4, dtype=torch.float64)
        for i in ids:
            f, s, _ = orc.refine(scene.make_window(5, 2, 40, seed=500 + i), opt)
            mine[i] = torch.from_numpy(f[0, 9:13].copy())
        dist.all_reduce(mine)  # disjoint rows: the sum is the gathered table
        q.put((rank, ok_moments, ok_count, ids, mine.numpy()))
    finally:
        dist.destroy_process_group()


def test_partition_plans_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(o[1] and o[2] for o in out)
    assert sorted(out[0][3] + out[1][3]) == list(range(7))          # every window exactly once
    assert np.array_equal(out[0][4], out[1][4])                      # same table on every rank
    import mvo_oracle_py as orc

    opt = orc.make_options(max_num_iterations=5, function_tolerance=0.0, gradient_tolerance=0.0,
                           parameter_tolerance=0.0)
    f, _, _ = orc.refine(scene.make_window(5, 2, 40, seed=503), opt)
    assert np.array_equal(out[0][4][3], f[0, 9:13])


def test_shard_ranges_tile_exactly():
    for n in (0, 1, 5, 64, 1001):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = sharding.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                covered += list(range(lo, hi))
            assert covered == list(range(n))
    w = scene.make_window(8, 3, 50, seed=3, span=("anchored", 2, 8))
    parts = [sharding.shard_observations(w, r, 3) for r in range(3)]
    assert sum(p.n_residual_blocks for p in parts) == w.n_residual_blocks
    for p in range(w.n_pairs):
        lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
        got = np.concatenate([q.m0[int(q.pair_offsets[p]):int(q.pair_offsets[p + 1])] for q in parts])
        assert np.array_equal(got, w.m0[lo:hi])


def test_landmark_shards_tile_the_scene():
    """Bundle-adjustment partition plan (include/mvo_hip_ba.h, partial_*): contiguous landmark ranges,
    balanced by observations, every observation exactly once, poses replicated."""
    from mvo_amd import ba_scene

    sc = ba_scene.make_ba_scene(12, 6, 257, seed=4, span=7)
    for world in (1, 2, 3, 8):
        parts = [sharding.shard_landmarks(sc, r, world) for r in range(world)]
        assert parts[0][1][0] == 0 and parts[-1][1][1] == sc.n_landmarks
        for (a, ra), (b, rb) in zip(parts, parts[1:]):
            assert ra[1] == rb[0]
        assert sum(p.n_observations for p, _ in parts) == sc.n_observations
        assert np.array_equal(np.concatenate([p.obs_ray for p, _ in parts]), sc.obs_ray)
        assert np.array_equal(np.concatenate([p.X for p, _ in parts]), sc.X)
        for p, _ in parts:
            assert p.poses is sc.poses and int(p.lm_offsets[0]) == 0
            assert abs(p.n_observations - sc.n_observations / world) <= 7


def test_landmark_shards_are_never_empty():
    """mvo_hip_ba_upload refuses L == 0, and a rank that raises while the others enter the all-reduce
    hangs the job: every rank owns at least one landmark however skewed the track lengths are, and a
    world larger than the scene is refused on every rank alike, before any collective."""
    from dataclasses import replace
    from mvo_amd import ba_scene

    sc = ba_scene.make_ba_scene(12, 6, 40, seed=5, span=7)
    # one very long track up front: an observation-balanced cut alone would leave ranks empty
    offs = np.asarray(sc.lm_offsets, dtype=np.int64)
    small = replace(sc, X=sc.X[:5], lm_offsets=offs[:6].astype(np.uint64), obs_frame=sc.obs_frame[:offs[5]],
                    obs_ray=sc.obs_ray[:offs[5]], true_X=sc.true_X[:5])
    for world in (2, 4, 5):
        parts = [sharding.shard_landmarks(small, r, world) for r in range(world)]
        assert all(p.n_landmarks >= 1 for p, _ in parts)
        assert sum(p.n_landmarks for p, _ in parts) == 5
        assert sum(p.n_observations for p, _ in parts) == small.n_observations
    for rank in range(8):
        with pytest.raises(ValueError):
            sharding.shard_landmarks(small, rank, 8)

"""bench.py's workload plan (CPU): which BASELINE.json configuration a run measures at N GPUs and how
the work is spread -- the N > 1 default and --workload config3 shard the SAME 4096 windows
(configs[2], strong scaling), config4 is partitioned inside the library, --weak keeps a copy per rank."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_plan_follows_baseline_configs():
    assert bench.plan(None, 1) == ("config2", "single")              # configs[1]: the metric's own configuration
    for n in (2, 4, 8):
        assert bench.plan(None, n) == ("config3", "strong_windows")  # configs[2]: 4096 windows sharded over 1/2/4/8
        assert bench.plan("config3", n) == ("config3", "strong_windows")
        assert bench.plan("config4", n) == ("config4", "partitioned")  # configs[3]
        assert bench.plan("config4r", n) == ("config4r", "partitioned")
        assert bench.plan("config3", n, weak=True) == ("config3", "weak")
        assert bench.plan("config2", n) == ("config2", "weak")
    assert bench.plan("config3", 1) == ("config3", "single")


def test_strong_window_shards_tile_the_same_4096_windows(monkeypatch):
    monkeypatch.setenv("MVO_BENCH_WINDOWS", "24")
    whole, _, _ = bench.build_workload("config3", 0, 1, "single")
    for world in (2, 3, 8):
        parts = [bench.build_workload("config3", r, world, "strong_windows")[0] for r in range(world)]
        assert sum(len(p) for p in parts) == 24
        flat = [w for p in parts for w in p]
        for a, b in zip(flat, whole):                                # same windows, same order, each exactly once
            assert np.array_equal(a.frames, b.frames) and np.array_equal(a.m0, b.m0)
    weak = [bench.build_workload("config3", r, 2, "weak")[0] for r in range(2)]
    assert len(weak[0]) == len(weak[1]) == 24 and not np.array_equal(weak[0][0].m0, weak[1][0].m0)


def test_traffic_figures_go_stale_with_the_kernel_source(tmp_path, monkeypatch):
    import json

    from mvo_amd.srchash import source_hash

    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = {"per_workload": {"config4": {"k_fold<false, true>": {"fetch_bytes": 100, "write_bytes": 5,
                                                                   "source_hash": source_hash("k_fold<false, true>")},
                                        "k_pass<true>": {"fetch_bytes": 7, "write_bytes": 1, "source_hash": "0123"}}}}
    (prof / "traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.measured_traffic("config4", "k_fold<false, true>") == 105
    assert bench.measured_traffic("config4", "k_pass<true>") is None      # measured on another source: stale
    assert bench.measured_traffic("config2", "k_fold<false, true>") is None


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE (the round driver's command) spawns the two ranks itself and
    relays their exit code.  Without a GPU every rank stops at "bench.py needs a GPU" -- which shows that the
    ranks were started with WORLD_SIZE = 2 (a lone process would have got past that check's precondition
    differently: `--gpus 2 but WORLD_SIZE=1`), and that the parent neither hides the failure nor touches torch."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: this test is about the launcher, not the GPU
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "starting 2 ranks" in out.stderr and "torch.distributed.run" in out.stderr
    assert "bench.py needs a GPU" in out.stderr
    assert "WORLD_SIZE=1" not in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # no line was faked

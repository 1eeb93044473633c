"""bench.py's N > 1 path on a one-GPU box: two ranks that share GPU 0 (MVO_BENCH_REHEARSE=1; gloo for the
bench's bookkeeping collectives, RCCL needs a GPU per rank).  Checks that the same config-3 windows are
sharded over the ranks, that rank 0 prints ONE well-formed line, and that the line says it is a rehearsal."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_shard_the_same_windows():
    env = dict(os.environ, MVO_BENCH_REHEARSE="1", MVO_BENCH_WINDOWS="256", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    # no launcher in front: `python bench.py --gpus 2` starts its two ranks itself (the driver's command)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "lm_iterations_per_sec" and d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["windows_whole_job"] == 256 and d["config"]["windows_per_gpu"] == 128
    assert "REHEARSAL" in d["data"] and d["value"] > 0 and d["steps"] == 3
    # the line carries its own efficiency baseline: all 256 windows on one GPU, and value / (that * N)
    one = d["same_workload_one_gpu"]
    assert one["windows"] == 256 and one["value"] > 0
    assert abs(d["efficiency_vs_one_gpu"] - d["value"] / (one["value"] * 2)) < 1e-3

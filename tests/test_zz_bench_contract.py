"""bench.py's one JSON line: the keys and types the round driver reads (metric of BASELINE.json on
configs[1], roofline of the assembly kernel, the CPU oracle timed beside it).  Collected last
(test_zz_...): it is the one test that depends on timing.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_default_bench_line_has_the_contract_fields():
    env = dict(os.environ)
    env.pop("MVO_HIP_POISON_ALLOC", None)  # measure what ships, not the test hygiene build of it
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly ONE line on stdout"
    b = json.loads(lines[0])
    assert b["metric"] == "lm_iterations_per_sec" and b["unit"] == "LM iterations/s"
    assert b["n_gpus"] == 1 and b["steps"] == 8 and b["warmup"] == 2
    assert b["higher_is_better"] is True and b["scaling"] == "weak" and b["vs_baseline"] is None
    assert b["dtype"] == "f64" and b["data"] == "synthetic"
    assert b["config"]["workload"].startswith("config2") and "model" not in b["config"]
    assert b["value"] > 0 and b["ms_per_step"] > 0
    # value = LM iterations of all windows / time: 10 iterations per step here
    assert abs(b["value"] - 10.0 / (b["ms_per_step"] * 1e-3)) <= 0.02 * b["value"]
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["bytes_per_launch"]
    big = b["roofline_large"]
    # measured 0.58-0.67; the margin keeps a noisy box from failing the suite
    assert big["frac"] >= 0.45, "the assembly kernel at config-4 size stays around half of the HBM peak or above"
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == b["unit"] and c["sample"]
    assert b["value"] > 100 * c["value"]

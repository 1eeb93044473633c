"""mvo_hip_comm_* / mvo_hip_solve_partitioned: the RCCL all-reduce inside the library, on the handle's
stream.  One GPU per box here, so the collective runs in a world of one -- through the C ABI in a
child process that never imports torch (system HIP runtime + system librccl.so.1, the reference's
setting).  The multi-rank logic (shards tile the work, sums are additive, every rank takes the same
step) is covered by tests/test_distributed_cpu.py (gloo) and tests/test_gpu_partitioned.py (two
handles as two ranks); a curve over 2-8 GPUs needs a node and is the driver's to measure."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from _util import FIXED10, refined
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_world_of_one_rccl_solve_through_the_c_abi_without_torch(oracle):
    env = dict(os.environ)
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py")], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for name in ("plain", "cauchy", "batch"):
        o = out[name]
        assert o["repeat_equal"]
        assert o["diff_single"] < 1e-9, (name, o["diff_single"])       # same sums, other summation route
        assert o["iterations"] == o["iterations_single"]
        assert np.allclose(o["final_cost"], o["final_cost_single"], rtol=1e-9, atol=0)
    assert out["reinit_iterations"] == 10
    assert out["after_abort"] == [capi.ERR_STATE, capi.ERR_STATE]  # mvo_hip_comm_abort: no communicator, no result
    assert out["after_abort_iterations"] == 10
    # and against the oracle
    w = scene.make_window(30, 5, 800, seed=41, span=("sliding", 18))
    fo, so, _ = oracle.refine(w, oracle.make_options(**FIXED10))
    assert np.abs(refined(np.array(out["plain"]["frames"])[0], 5) - refined(fo, 5)).max() < 1e-6
    w = scene.make_window(12, 4, 600, seed=43, outlier_fraction=0.2)
    fo, so, _ = oracle.refine(w, oracle.make_options(use_loss_function=True, **FIXED10))
    assert np.abs(refined(np.array(out["cauchy"]["frames"])[0], 4) - refined(fo, 4)).max() < 1e-6

"""Child process of tests/test_gpu_rccl.py: the observation-partitioned solve with the RCCL collective
inside libmvo_hip, through the C ABI alone -- this process never imports torch, so the library runs on
the system's HIP runtime and the system's librccl.so.1, as it would inside the reference's node."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "monocular-visual-odometry_amd"), os.path.dirname(os.path.abspath(__file__))]

import numpy as np  # noqa: E402

from mvo_amd import capi, scene  # noqa: E402

assert "torch" not in sys.modules
FIXED10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
out = {}
with capi.Solver(0, profiling=False) as s:
    assert capi.load().mvo_hip_solve_partitioned(s._h) == capi.ERR_STATE  # no upload, no communicator
    uid = capi.comm_unique_id()
    assert len(uid) == capi.COMM_ID_BYTES
    s.comm_init(uid, 0, 1)
    assert capi.load().mvo_hip_comm_init(s._h, uid, 0, 1) == capi.ERR_STATE  # one communicator per handle
    for name, w, kw in (("plain", scene.make_window(30, 5, 800, seed=41, span=("sliding", 18)), {}),
                        ("cauchy", scene.make_window(12, 4, 600, seed=43, outlier_fraction=0.2), dict(use_loss_function=True)),
                        ("batch", None, {})):
        ws = [scene.config3_window(i) for i in range(6)] if w is None else [w]
        opt = capi.make_options(**kw, **FIXED10)
        s.upload(ws, opt)
        s.solve_partitioned_rccl()
        fp, sp = s.download()
        s.solve_partitioned_rccl()          # twice in a row on the same upload
        fq, sq = s.download()
        s.solve()
        fs, ss = s.download()
        out[name] = dict(diff_single=float(np.abs(fp - fs).max()), repeat_equal=bool(np.array_equal(fp, fq)),
                         iterations=[int(x.iterations) for x in sp], iterations_single=[int(x.iterations) for x in ss],
                         final_cost=[float(x.final_cost) for x in sp], final_cost_single=[float(x.final_cost) for x in ss],
                         frames=fp.tolist())
    s.comm_destroy()
    s.comm_destroy()                        # idempotent
    assert capi.load().mvo_hip_solve_partitioned(s._h) == capi.ERR_STATE
    s.comm_init(capi.comm_unique_id(), 0, 1)  # and a new communicator on the same handle works
    s.upload([scene.make_window(6, 3, 120, seed=11)], capi.make_options(**FIXED10))
    s.solve_partitioned_rccl()
    out["reinit_iterations"] = int(s.download()[1][0].iterations)
    # the fail-stop exit: abort instead of destroy; the handle then holds neither communicator nor result
    s.comm_abort()
    s.comm_abort()                          # idempotent
    L = capi.load()
    out["after_abort"] = [int(L.mvo_hip_solve_partitioned(s._h)), int(L.mvo_hip_download(s._h, None, None))]
    s.comm_init(capi.comm_unique_id(), 0, 1)
    s.solve_partitioned_rccl()
    out["after_abort_iterations"] = int(s.download()[1][0].iterations)
print("RESULT " + json.dumps(out))

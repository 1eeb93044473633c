"""One window of config 2's shape by number of landmarks: fold and LM loop in one launch (k_lm<256, 7>) against two
launches (MVO_HIP_NO_FUSED), solve time without event records.  MVO_HIP_FUSED_MAX_BLOCKS moves the fused launch's limit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
import numpy as np
from mvo_amd import capi, scene
fixed = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
def handle(**env):
    os.environ.update(env)
    try:
        return capi.Solver(0, profiling=False)
    finally:
        for k in env: del os.environ[k]
a = handle(MVO_HIP_FUSED_MAX_BLOCKS="100000")
b = handle(MVO_HIP_NO_FUSED="1")
for L in (5000, 7000, 10000, 15000, 20000, 30000):
    w = scene.make_window(30, 5, L, seed=1002, span=("sliding", 20))
    row = []
    for name, s in (("fused", a), ("two launches", b)):
        s.upload([w], capi.make_options(**fixed))
        for _ in range(20): s.solve()
        ts = []
        for _ in range(200):
            t = time.perf_counter(); s.solve(); ts.append(time.perf_counter() - t)
        tm = s.timing()
        row.append("%s %.1f us (%d+%d)" % (name, np.median(ts) * 1e6, tm["assembly_launches"], tm["lm_launches"]))
    print("%d landmarks, %d blocks: %s" % (L, w.n_residual_blocks, " | ".join(row)), flush=True)
a.close(); b.close()

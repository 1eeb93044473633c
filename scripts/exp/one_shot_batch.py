"""mvo_hip_refine_batch (one-shot: upload, solve, download) of small batches whose rays fit the arena, with the fold reading
the rays in the pinned arena (default) and with the arena copied (MVO_HIP_ARENA_COPY)."""
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
a, b = handle(), handle(MVO_HIP_ARENA_COPY="1")
for W, L in ((1, 300), (1, 1500), (4, 300), (16, 100), (6, 300)):
    ws = [scene.make_window(10, 5, L, seed=40 + i) for i in range(W)]
    opt = capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **fixed)
    row = []
    for name, s in (("in place", a), ("copied", b), ("in place", a), ("copied", b)):
        for _ in range(10): s.refine_batch(ws, opt)
        ts = []
        for _ in range(100):
            t = time.perf_counter(); s.refine_batch(ws, opt); ts.append(time.perf_counter() - t)
            ts[-1] = s.timing()["upload_seconds"] + s.timing()["solve_seconds"]
        row.append("%s %.1f" % (name, np.median(ts) * 1e6))
    print("%d windows, %d blocks in all: upload + solve us | %s" % (W, sum(w.n_residual_blocks for w in ws), " | ".join(row)), flush=True)

"""One window under the Cauchy loss, by size: inline passes by the window's own workgroup (the default up to 8192 residual
blocks) against the tiled passes (MVO_HIP_ASSEMBLY_TILED: the persistent solver k_lm<256, 6>).  Solve time, upload once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
import numpy as np
from mvo_amd import capi, scene
fixed = dict(use_loss_function=True, max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
with capi.Solver(0, profiling=False) as s:
    shapes = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
    for N, M, L in shapes or ((3, 2, 60), (3, 2, 300), (5, 2, 200), (10, 5, 30), (10, 5, 60), (10, 5, 120), (10, 5, 230), (20, 5, 60), (30, 5, 60)):
        w = scene.make_window(N, M, L, seed=31)
        row = []
        for name, mode in (("inline", 0), ("tiled", capi.ASSEMBLY_TILED)):
            s.upload([w], capi.make_options(assembly_mode=mode, **fixed))
            for _ in range(30): s.solve()
            ts = []
            for _ in range(300):
                t = time.perf_counter(); s.solve(); ts.append(time.perf_counter() - t)
            tm = s.timing()
            row.append("%s %.1f us (%d+%d launches)" % (name, np.median(ts) * 1e6, tm["assembly_launches"], tm["lm_launches"]))
        print("N=%d M=%d %d features, %d blocks, %d pairs: %s" % (N, M, L, w.n_residual_blocks, w.n_pairs, " | ".join(row)), flush=True)

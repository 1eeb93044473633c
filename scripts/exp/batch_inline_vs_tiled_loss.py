"""Batches of windows below the inline limit (8192 residual blocks): every window folded by its own workgroup (inline, the
default) against the chip-wide fold + LM kernels reading its slots (MVO_HIP_ASSEMBLY_TILED).  Call time, no event records."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
import numpy as np
from mvo_amd import capi, scene
fixed = dict(use_loss_function=True, max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
with capi.Solver(0, profiling=False) as s:
    for L in [int(a) for a in sys.argv[1:]] or (30, 100, 200):
        base = [scene.make_window(10, 5, L, seed=900 + i) for i in range(16)]
        for W in (64, 128, 256, 512, 1024):
            ws = [base[i % 16] for i in range(W)]
            row = []
            for name, mode in (("inline", 0), ("tiled", capi.ASSEMBLY_TILED)):
                s.upload(ws, capi.make_options(assembly_mode=mode, **fixed))
                for _ in range(10): s.solve()
                ts = []
                for _ in range(60):
                    t = time.perf_counter(); s.solve(); ts.append(time.perf_counter() - t)
                row.append("%s %.1f us" % (name, np.median(ts) * 1e6))
            print("%d windows of %d blocks: %s" % (W, ws[0].n_residual_blocks, " | ".join(row)), flush=True)

"""Batches of config-3 windows: four windows per wavefront (k_lm4) against one window per wavefront / workgroup, by size
of the batch.  Prints the solve's device time (HIP events) minus the fold's, i.e. the LM part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi, scene
fixed = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
base = [scene.config3_window(i) for i in range(32)]
with capi.Solver(0) as s:
    sizes = [int(a) for a in sys.argv[1:]] or [260, 384, 512, 768, 1024, 1536, 2048, 4096]
    for W in sizes:
        ws = [base[i % 32] for i in range(W)]
        row = []
        for name, mode in (("auto", 0), ("four/wave", capi.ASSEMBLY_FOUR_PER_WAVE), ("one/wave", capi.ASSEMBLY_ONE_PER_WAVE)):
            s.upload(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | mode, **fixed))
            s.set_profiling(True)
            best = (1e9, 0, 0)
            for _ in range(5):
                s.solve()
                tm = s.timing()
                best = min(best, (tm["device_seconds"], tm["assembly_seconds"], tm["solve_seconds"]))
            row.append("%s: device %.1f us (fold %.1f), call %.1f us" % (name, best[0] * 1e6, best[1] * 1e6, best[2] * 1e6))
        print(W, "windows |", " | ".join(row), flush=True)

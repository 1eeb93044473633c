"""k_lm4 against k_lm<64> by batch size, CALL time of mvo_hip_solve without event records (config-3 windows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
import numpy as np
from mvo_amd import capi, scene
fixed = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
base = [scene.config3_window(i) for i in range(32)]
sizes = [int(a) for a in sys.argv[1:]] or [320, 384, 448, 512, 576, 640, 704, 768, 1024]
with capi.Solver(0, profiling=False) as s:
    for W in sizes:
        ws = [base[i % 32] for i in range(W)]
        row = []
        for name, mode in (("auto", 0), ("four/wave", capi.ASSEMBLY_FOUR_PER_WAVE), ("one/wave", capi.ASSEMBLY_ONE_PER_WAVE)):
            s.upload(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | mode, **fixed))
            for _ in range(10): s.solve()
            ts = []
            for _ in range(100):
                t = time.perf_counter(); s.solve(); ts.append(time.perf_counter() - t)
            row.append("%s %.1f" % (name, np.median(ts) * 1e6))
        print(W, "windows, call us |", " | ".join(row), flush=True)

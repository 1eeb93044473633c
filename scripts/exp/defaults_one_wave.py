"""The reference's shipped defaults (N=3, M=2, 60 features) through the four-wavefront kernel (k_lm<256>, the default
up to 256 windows) and through the one-wavefront kernel (k_lm<64>, MVO_HIP_ASSEMBLY_ONE_PER_WAVE): solve time and
device time of one window."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
import numpy as np
from mvo_amd import capi, scene
w = scene.reference_defaults_window()
for kw in ({}, dict(use_loss_function=True)):
    for name, mode in (("k_lm<256>", 0), ("k_lm<64>", capi.ASSEMBLY_ONE_PER_WAVE)):
        for prof in (False, True):
            with capi.Solver(0, profiling=prof) as s:
                opt = capi.make_options(assembly_mode=mode, **kw)
                for _ in range(50): s.refine(w, opt)
                ts, dv = [], []
                for _ in range(400):
                    t = time.perf_counter(); f, sm = s.refine(w, opt); ts.append(time.perf_counter() - t)
                    dv.append(s.timing()["device_seconds"])
                print(f"{kw} {name} events {'on ' if prof else 'off'}: refine() median {np.median(ts)*1e6:.1f} us, device {np.median(dv)*1e6:.1f} us, "
                      f"iterations {sm.iterations} cost {sm.final_cost:.6e}", flush=True)

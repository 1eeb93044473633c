"""End-to-end latency of mvo_hip_refine (host buffers in, host buffers out) at the reference's
shipped defaults (numberToNote=3, numberToRefine=2, <=60 features, tolerances 1e-6/1e-10/1e-8,
50 iterations max) -- the call Refiner::stage makes once per keyframe."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from mvo_amd import capi, scene
import mvo_oracle_py as orc

for label, w, kw in (("defaults N=3 M=2 60 feat", scene.reference_defaults_window(), {}),
                     ("defaults + Cauchy", scene.reference_defaults_window(), dict(use_loss_function=True)),
                     ("N=10 M=5 300 feat", scene.config1(), {})):
  for prof in (False, True):
    with capi.Solver(0, profiling=prof) as s:
        opt = capi.make_options(**kw)
        for _ in range(20): s.refine(w, opt)
        ts = []
        for _ in range(200):
            t = time.perf_counter(); f, sm = s.refine(w, opt); ts.append(time.perf_counter() - t)
        ts = np.array(ts) * 1e6
        t0 = time.perf_counter()
        for _ in range(20): fo, so, _ = orc.refine(w, orc.make_options(**kw))
        cpu = (time.perf_counter() - t0) / 20 * 1e6
        print(f"{label} [events {'on' if prof else 'off'}]: refine() median {np.median(ts):.1f} us, p90 {np.percentile(ts, 90):.1f} us, iterations {sm.iterations}; "
              f"timing {s.timing()['upload_seconds']*1e6:.1f} us upload + {s.timing()['solve_seconds']*1e6:.1f} us solve "
              f"(device {s.timing()['device_seconds']*1e6:.1f} us); CPU oracle {cpu:.1f} us")

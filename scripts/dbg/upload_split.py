import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
import numpy as np
from mvo_amd import capi, scene
w = scene.config2(seed=1002)
opt = capi.make_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
with capi.Solver(0) as s:
    for _ in range(10): s.refine(w, opt)
    ups, sols, tot = [], [], []
    for _ in range(100):
        t = time.perf_counter(); s.refine(w, opt); tot.append(time.perf_counter() - t)
        tm = s.timing(); ups.append(tm["upload_seconds"]); sols.append(tm["solve_seconds"])
    print("refine median %.1f us; upload %.1f us; solve %.1f us" % (np.median(tot)*1e6, np.median(ups)*1e6, np.median(sols)*1e6))
    # upload alone, then solve alone
    ups2 = []
    for _ in range(50):
        t = time.perf_counter(); s.upload([w], opt); ups2.append(time.perf_counter() - t); s.solve(); s.download()
    print("upload() call median %.1f us" % (np.median(ups2)*1e6))

"""33 windows, fixed 10 iterations: k_lm4 against the one-window-per-workgroup path and the CPU oracle, per window."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mvo_amd import capi, scene
import mvo_oracle_py as orc
orc.lib()
opts = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
ws = [scene.make_window(10, 5, 120, seed=300 + i) for i in range(33)]
with capi.Solver(0) as s:
    f4, s4 = s.refine_batch(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_FOUR_PER_WAVE, **opts))
    f1, s1 = s.refine_batch(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_ONE_PER_WAVE, **opts))
f4, f1 = np.asarray(f4), np.asarray(f1)
for i, w in enumerate(ws):
    fo, so, _ = orc.refine(w, orc.make_options(**opts))
    fo = np.asarray(fo)
    d41 = np.abs(f4[i] - f1[i]).max(); d4o = np.abs(f4[i][:5, 9:13] - fo[:5, 9:13]).max(); d1o = np.abs(f1[i][:5, 9:13] - fo[:5, 9:13]).max()
    if d41 > 1e-9 or d4o > 1e-8:
        print(i, "k_lm4 - one/wave %.2e   k_lm4 - oracle %.2e   one/wave - oracle %.2e   costs %.17g %.17g %.17g" % (d41, d4o, d1o, s4[i].final_cost, s1[i].final_cost, so.final_cost))
print("max k_lm4 - one/wave %.3e" % np.abs(f4 - f1).max())

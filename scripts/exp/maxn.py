import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("monocular-visual-odometry_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from mvo_amd import capi, scene
import mvo_oracle_py as orc
from _util import FIXED10, refined
w = scene.make_window(256, 5, 1500, seed=99, span=("sliding", 40))
s = capi.Solver(0)
s.set_trace(64)
for mode in (capi.ASSEMBLY_AUTO, capi.ASSEMBLY_PASSES):
    fg, sg = s.refine(w, capi.make_options(assembly_mode=mode, **FIXED10))
    print("mode", mode, "GPU: term", sg.termination, "reason", sg.reason, "iters", sg.iterations, "cost %.12e -> %.12e" % (sg.initial_cost, sg.final_cost))
    for i, t in enumerate(s.trace()):
        print("   it", i, "cost %.12e change %.3e gmax %.3e step %.3e rho %.4f radius %.3e ok %d" % tuple((t[k] if isinstance(t, dict) else getattr(t, k)) for k in ("cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius", "step_is_successful")))
fo, so, tr = orc.refine(w, orc.make_options(**FIXED10))
print("oracle: term", so.termination, "reason", so.reason, "iters", so.iterations, "cost %.12e -> %.12e" % (so.initial_cost, so.final_cost))
for i, t in enumerate(tr):
    print("   it", i, "cost %.12e change %.3e gmax %.3e step %.3e rho %.4f radius %.3e ok %d" % tuple((t[k] if isinstance(t, dict) else getattr(t, k)) for k in ("cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius", "step_is_successful")))
print("max diff refined", np.abs(refined(fg, 5) - refined(fo, 5)).max())

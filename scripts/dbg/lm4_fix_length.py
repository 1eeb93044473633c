"""k_lm4 against the one-window-per-wavefront path and the CPU oracle with fixLength and zero tolerances: where do the
solves end, and at what cost (the end is the first evaluation that repeats the cost bit for bit)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mvo_amd import capi, scene
import mvo_oracle_py as orc
orc.lib()
opts = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0, fix_length=True)
ws = [scene.make_window(10, 5, 120, seed=300 + i) for i in range(33)]
with capi.Solver(0) as s:
    f4, s4 = s.refine_batch(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_FOUR_PER_WAVE, **opts))
    f1, s1 = s.refine_batch(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED | capi.ASSEMBLY_ONE_PER_WAVE, **opts))
for i, (w, a, b) in enumerate(zip(ws, s4, s1)):
    fo, so, tr = orc.refine(w, orc.make_options(**opts))
    key = lambda x: (x.termination, x.reason, x.iterations, x.num_successful_steps)
    flag = "" if key(a) == key(b) else "   <-- differ"
    print(i, "k_lm4", key(a), "%.17g" % a.final_cost, "| one/wave", key(b), "%.17g" % b.final_cost, "| oracle", key(so), "%.17g" % so.final_cost, flag)

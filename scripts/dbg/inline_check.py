"""debug: the sequence of solves tests/test_gpu_parity.py runs on one handle, GPU vs oracle initial / final cost"""
import os, sys
os.environ.setdefault("MVO_HIP_POISON_ALLOC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "monocular-visual-odometry_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
from mvo_amd import capi, scene
import mvo_oracle_py as orc
FIXED10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
shapes = {
    "defaults_n3m2": lambda: scene.reference_defaults_window(),
    "minimal_n2m1": lambda: scene.make_window(2, 1, 50, seed=21),
    "n6m3": lambda: scene.make_window(6, 3, 120, seed=11),
    "n6m5": lambda: scene.make_window(6, 5, 70, seed=12),
    "config1": lambda: scene.config1(),
    "ragged": lambda: scene.make_window(12, 4, 200, seed=13, span=("anchored", 2, 12)),
    "n40m5_sliding": lambda: scene.make_window(40, 5, 600, seed=14, span=("sliding", 15)),
}
with capi.Solver(0) as s:
    for mode in (0, 1, 2, 3):
        for name, mk in shapes.items():
            w = mk()
            capi.poison_lds(0)
            f, sm = s.refine(w, capi.make_options(assembly_mode=mode, **FIXED10))
            fo, so, _ = orc.refine(w, orc.make_options(**FIXED10))
            cnt = np.diff(w.pair_offsets)
            bad = abs(sm.final_cost - so.final_cost) > 1e-8 * so.initial_cost or abs(sm.initial_cost - so.initial_cost) > 1e-9 * so.initial_cost
            print(name, "mode", mode, "obs", int(cnt.sum()), "empty pairs", int((cnt == 0).sum()), "initial", sm.initial_cost, so.initial_cost, "final", sm.final_cost, so.final_cost, "BAD" if bad else "ok")

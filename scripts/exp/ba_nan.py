import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("monocular-visual-odometry_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa
from mvo_amd import ba_capi, ba_scene
import mvo_ba_oracle_py as bo
sc = ba_scene.make_ba_scene(8, 5, 120, seed=3, span=6)
sc.obs_ray[17, 1] = np.nan
kw = dict(max_num_iterations=8, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
with ba_capi.BaSolver(0) as ba:
    poses, X, s = ba.refine(sc, ba_capi.make_options(**kw))
    print("GPU: term", s.termination, "iters", s.iterations, "cost", s.initial_cost, s.final_cost, "poses finite", np.isfinite(poses).all(), "X finite", np.isfinite(X).all(),
          "poses unchanged", np.array_equal(poses, sc.poses))
po, Xo, so, tr = bo.refine(sc, bo.make_options(**kw))
print("oracle: term", so.termination, "iters", so.iterations, "cost", so.initial_cost, so.final_cost, "poses finite", np.isfinite(po).all(), "X finite", np.isfinite(Xo).all(),
      "poses unchanged", np.array_equal(po, sc.poses))

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "monocular-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch  # noqa
import numpy as np
from mvo_amd import capi, scene
import mvo_oracle_py as orc
from _util import FIXED10
LONG = dict(max_num_iterations=40, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
import numpy as np
def rank_one(v0, N, M, n=8):
    frames = np.zeros((N, 13)); frames[:, 0] = frames[:, 4] = frames[:, 8] = 1.0
    frames[:, 9:12] = np.asarray(v0) / np.linalg.norm(v0); frames[:, 12] = 1.0
    P = scene.num_pairs(M, N)
    return scene.Window(N=N, M=M, frames=frames, pair_offsets=np.arange(P + 1, dtype=np.uint64) * n,
                        m0=np.tile([0.0, 1.0, 0.0], (n * P, 1)), m1=np.tile([0.0, 0.0, 1.0], (n * P, 1)),
                        true_vec=np.zeros((N, 3)), true_scale=np.zeros(N))
Z60 = dict(max_num_iterations=60, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
cases = [
    ("z0 n10m5 loss fix tiled", scene.make_window(10, 5, 150, seed=955, direction=("z", 0.0), outlier_fraction=0.2), dict(use_loss_function=True, fix_length=True, **FIXED10), capi.ASSEMBLY_TILED),
    ("rank one 2,1", rank_one([0.1, 0.3, 0.95], 2, 1), Z60, 0),
]
s = capi.Solver(0)
for name, w, kw, mode in cases:
    fo, so, to = orc.refine(w, orc.make_options(**kw))
    s.set_trace(64)
    fg, sg = s.refine(w, capi.make_options(assembly_mode=mode, **kw))
    tg = s.trace(64)
    print("==", name, "gpu", (sg.termination, sg.reason, sg.iterations), "oracle", (so.termination, so.reason, so.iterations))
    for i in range(max(len(tg), len(to))):
        a = tg[i] if i < len(tg) else None
        b = to[i] if i < len(to) else None
        f = lambda r: "-" if r is None else "%d%d c=%.17e dc=%.3e g=%.3e st=%.3e rd=%.3e r=%.3e" % (r["step_is_valid"], r["step_is_successful"], r["cost"], r["cost_change"], r["gradient_max_norm"], r["step_norm"], r["relative_decrease"], r["trust_region_radius"])
        print(i, "G", f(a))
        print(i, "O", f(b))
    print("gpu refined", fg[:w.M, 9:13])
    print("ora refined", fo[:w.M, 9:13])

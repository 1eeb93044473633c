import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "monocular-visual-odometry_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch  # noqa
import numpy as np
from mvo_amd import capi, scene
import mvo_oracle_py as orc
def rank_one(v0, N, M, n=8):
    frames = np.zeros((N, 13)); frames[:, 0] = frames[:, 4] = frames[:, 8] = 1.0
    frames[:, 9:12] = np.asarray(v0) / np.linalg.norm(v0); frames[:, 12] = 1.0
    P = scene.num_pairs(M, N)
    return scene.Window(N=N, M=M, frames=frames, pair_offsets=np.arange(P + 1, dtype=np.uint64) * n,
                        m0=np.tile([0.0, 1.0, 0.0], (n * P, 1)), m1=np.tile([0.0, 0.0, 1.0], (n * P, 1)),
                        true_vec=np.zeros((N, 3)), true_scale=np.zeros(N))
Z60 = dict(max_num_iterations=60, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
s = capi.Solver(0)
rng = np.random.default_rng(5)
for i in range(24):
    v0 = [rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6), 0.8]
    N, M = [(2, 1), (3, 2), (3, 1)][i % 3]
    n = [8, 1, 3, 50][i % 4]
    w = rank_one(v0, N, M, n)
    for mode in (0, 2):
        fg, sg = s.refine(w, capi.make_options(assembly_mode=mode, **Z60))
        fo, so, _ = orc.refine(w, orc.make_options(**Z60))
        print(i, N, M, n, mode, "gpu", (sg.termination, sg.reason, sg.iterations, sg.num_successful_steps), "%.3e" % sg.final_cost,
              "oracle", (so.termination, so.reason, so.iterations, so.num_successful_steps), "back==init", np.array_equal(fg[:M, 9:12], w.frames[:M, 9:12]))

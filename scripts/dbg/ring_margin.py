"""Development: margin of test_refine_sequence_on_a_sliding_ring_with_changing_shapes (fused track path vs
host-gathered solve) per step, for the library MVO_HIP_LIB_PATH points at."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi, scene, tracks_capi
w = scene.make_window(40, 5, 350, seed=4888, span=("sliding", 25), keep_tracks=True)
frames = scene.track_frames(w)
width = max(len(p) for _, p in frames)
opt = capi.make_options(max_num_iterations=6, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
shapes = [(3, 8), (5, 30), (2, 5), (5, 22), (1, 2), (4, 30)]
with capi.Solver(0) as solver, tracks_capi.TrackTable(solver, max_frames=32, max_features=width) as ring:
    for k, (rays, pre) in enumerate(frames):
        ring.push(rays, pre)
        M, N = shapes[k % len(shapes)]
        if k + 1 < N:
            continue
        fr = w.frames[w.N - 1 - k: w.N - 1 - k + N].copy()
        got, gs = ring.refine(fr, M, opt)
        offs, m0, m1 = ring.gather(M, N)
        w2 = scene.Window(N=N, M=M, frames=fr, pair_offsets=offs, m0=m0, m1=m1, true_vec=w.true_vec[:N],
                          true_scale=w.true_scale[:N], bearing_observations=0, meta={})
        exp, es = solver.refine(w2, opt)
        print(k, (M, N), "diff %.2e" % np.abs(got - exp).max(), "cost", es.final_cost, gs.final_cost, "it", gs.iterations, es.iterations)

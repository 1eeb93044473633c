"""config1, no loss: host-gathered route vs track-table route, with and without MVO_HIP_POISON_ALLOC."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
if len(sys.argv) > 1:
    from mvo_amd import capi, scene, tracks_capi
    N, M, L, span = 10, 5, 300, "full"
    w = scene.make_window(N, M, L, seed=4200 + N, span=span, keep_tracks=True)
    frames = scene.track_frames(w)
    opt = capi.make_options(max_num_iterations=15, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    s = capi.Solver(0)
    with tracks_capi.TrackTable(s, max_frames=N + 3, max_features=max(len(p) for _, p in frames)) as table:
        for rays, pre in frames:
            table.push(rays, pre)
        got, gs = table.refine(w.frames, M, opt)
        offs, m0, m1 = table.gather(M, N)
    w2 = scene.Window(N=N, M=M, frames=w.frames, pair_offsets=offs, m0=m0, m1=m1, true_vec=w.true_vec,
                      true_scale=w.true_scale, bearing_observations=w.bearing_observations, meta={})
    exp, es = s.refine(w2, opt)
    exp2, _ = s.refine(w2, opt)
    np.savez(sys.argv[1], got=got, exp=exp, exp2=exp2)
    sys.exit(0)
out = {}
for tag, env in (("plain", {}), ("poison", {"MVO_HIP_POISON_ALLOC": "1"})):
    f = os.path.join(ROOT, "gpurun_out", f"routes_{tag}.npz")
    e = dict(os.environ); e.pop("MVO_HIP_POISON_ALLOC", None); e.update(env)
    subprocess.check_call([sys.executable, __file__, f], env=e)
    out[tag] = np.load(f)
base = out["plain"]["exp"]
for tag in out:
    for k in ("got", "exp", "exp2"):
        d = np.abs(out[tag][k] - base)
        print(tag, k, "max |diff| vs plain host route:", d.max(), "entries differing:", int((d > 0).sum()))

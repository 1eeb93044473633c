"""How much of the fold's bandwidth do SHORT tiles cost?  Batches of 4096 windows with the same number of residual
blocks (10.5 k per window, 2.06 GB of rays in all) cut into 35, 17 or 9 frame pairs per window: 300, 618 or 1167
blocks per tile.  Prints the fold's duration by HIP events and the rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi, scene
fixed = dict(max_num_iterations=2, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for N, M, feats in ((10, 5, 300), (10, 2, 618), (10, 1, 1167)):
    base = [scene.make_window(N, M, feats, seed=50 + i) for i in range(8)]
    ws = [base[i % 8] for i in range(W)]
    blocks = sum(w.n_residual_blocks for w in ws)
    with capi.Solver(0) as s:
        s.upload(ws, capi.make_options(assembly_mode=capi.ASSEMBLY_TILED, **fixed))
        s.set_profiling(True)
        best = 1e9
        for _ in range(6):
            s.solve()
            tm = s.timing()
            best = min(best, tm["assembly_seconds"] / max(1, tm["assembly_launches"]))
        print(f"N={N} M={M} {feats} features: {ws[0].n_pairs} pairs/window, {blocks / W / ws[0].n_pairs:.0f} blocks/tile, "
              f"{48 * blocks / 1e9:.2f} GB: fold {best * 1e6:.1f} us = {48 * blocks / best / 1e12:.2f} TB/s", flush=True)

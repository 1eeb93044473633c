"""Development: the persistent robust solver (k_lm<256, 6>) solved over and over -- every result must equal the first
one bit for bit (a stale device-scope read or a barrier passed too early would show as a different trajectory), and
every solve must have stayed on the persistent path (lm_launches == 1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi, scene
FIXED10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for label, w in (("config5", scene.config2(seed=1005, outlier_fraction=0.2)),
                 ("N=60 M=4 900 landmarks", scene.make_window(60, 4, 900, seed=4302, outlier_fraction=0.2, span=("sliding", 20)))):
    with capi.Solver(0) as s:
        s.upload([w], capi.make_options(use_loss_function=True, **FIXED10))
        s.solve()
        ref, sr = s.download()
        ref = ref.copy()
        bad = fell_back = 0
        t0 = time.perf_counter()
        for i in range(reps):
            s.solve()
            f, sm = s.download(copy=False)
            if not np.array_equal(f, ref) or sm[0].final_cost != sr[0].final_cost:
                bad += 1
            if i % 500 == 0 and s.timing()["lm_launches"] != 1:
                fell_back += 1
        dt = time.perf_counter() - t0
        print(f"{label}: {reps} solves, {bad} differ from the first, fell back {fell_back}, {dt / reps * 1e3:.4f} ms per solve", flush=True)
        assert bad == 0 and fell_back == 0

"""Diagnostic: where an LM iteration spends its cycles (needs libmvo_hip_stamps.so, built with
`make -C monocular-visual-odometry_amd libmvo_hip_stamps.so`).  Shares, not absolute times."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi, scene
capi.LIB_PATH = os.path.join(ROOT, "monocular-visual-odometry_amd", "libmvo_hip_stamps.so")
lib = capi.load()
lib.mvo_hip_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
names = ["iterations", "per-pair", "w0 behind B2", "W/y", "blocks", "lm_advance", "prologue", "epilogue", "adv:decide", "adv:finalize", "adv:setup", "adv:eliminate", "adv:mcc", "adv:plus", "adv:publish", "adv:take_z"]
fixed = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
for label, ws in (("defaults N=3 M=2", [scene.reference_defaults_window()]), ("config1", [scene.config1()]),
                  ("config2", [scene.config2()]), ("config4", [scene.config4()]), ("config1 x 64", [scene.config3_window(i) for i in range(64)])):
    with capi.Solver(0) as s:
        s.upload(ws, capi.make_options(**fixed))
        s.solve()
        buf = (C.c_ulonglong * 16)()
        lib.mvo_hip_debug_stamps(buf, 1)
        s.solve()
        lib.mvo_hip_debug_stamps(buf, 0)
        n = max(1, buf[0])
        print(label, "evaluations", buf[0], " cycles/eval:", {names[i]: buf[i] // n for i in range(1, 6)}, "total", sum(buf[1:6]) // n, " per LAUNCH: prologue", buf[6], "epilogue", buf[7])
        print("    inside lm_advance:", {names[i]: buf[i] // n for i in range(8, 16)})

# the robust path: one evaluation per k_lm launch -- prologue / epilogue cycles per LAUNCH beside the evaluation's
for label, ws in (("config5 (Cauchy, tiled passes)", [scene.config2(seed=1005, outlier_fraction=0.2)]),
                  ("config4r", [scene.config4(seed=1004, outlier_fraction=0.2)])):
    with capi.Solver(0) as s:
        s.upload(ws, capi.make_options(use_loss_function=True, **fixed))
        s.solve()
        buf = (C.c_ulonglong * 16)()
        lib.mvo_hip_debug_stamps(buf, 1)
        s.solve()
        lib.mvo_hip_debug_stamps(buf, 0)
        n = max(1, buf[0])
        print(label, "evaluations", buf[0], " cycles/eval:", {names[i]: buf[i] // n for i in range(1, 8)})
        rb = (C.c_ulonglong * 8)()
        lib.mvo_hip_debug_stamps_persistent.argtypes = [C.POINTER(C.c_ulonglong)]
        lib.mvo_hip_debug_stamps_persistent(rb)
        if rb[0]:  # one window whose pass grid fits the chip: the persistent robust solver (k_lm<256, 6>), per pass
            rn = ["evaluation + step + unit baselines", "pass over the own tiles", "slots out", "tickets + finished pairs projected", "grid barrier"]
            print("    persistent solver, cycles per pass (workgroup 0):", {rn[i]: rb[i] // max(1, n - 1) for i in range(5)})

"""Diagnostic: cycles of k_ba_solve's phases (needs libmvo_hip_stamps.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "monocular-visual-odometry_amd", "libmvo_hip_stamps.so")
from mvo_amd import ba_capi, ba_scene
lib = ba_capi.load()
lib.mvo_hip_ba_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
names = ["load", "diag", "panel", "update", "backsub", "tail", "dump"]
for label, sc in (("ba2", ba_scene.ba_config2()), ("ba4", ba_scene.ba_config4())):
  with ba_capi.BaSolver(0) as s:
    s.upload(sc, ba_capi.make_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
    s.solve()
    buf = (C.c_ulonglong * 16)()
    lib.mvo_hip_ba_debug_stamps(buf, 1)
    s.solve()
    lib.mvo_hip_ba_debug_stamps(buf, 0)
    print(label, "k_ba_solve:", {n: buf[i] // 10 for i, n in enumerate(names)}, "cycles per call; total", sum(buf[:7]) // 10)
    print(label, "k_ba_schur (workgroup 0, thread 0):", {n: buf[8 + i] // 10 for i, n in enumerate(["phase A", "wait A", "phase B", "wait B"])}, "cycles per call")

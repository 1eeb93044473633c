"""Diagnostic: where k_lm4 (four windows per wavefront, batches of small windows) spends its cycles.  Needs
libmvo_hip_stamps.so (`make -C monocular-visual-odometry_amd libmvo_hip_stamps.so`).  Workgroup 0 only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import capi, scene
capi.LIB_PATH = os.path.join(ROOT, "monocular-visual-odometry_amd", "libmvo_hip_stamps.so")
lib = capi.load()
lib.mvo_hip_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.mvo_hip_debug_stamps_lm4.argtypes = [C.POINTER(C.c_ulonglong)]
names = ["turns", "per-pair", "tables", "loop exit", "blocks+gradient", "advance", "prologue", "epilogue"]
fixed = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ws = [scene.config3_window(i) for i in range(W)]
with capi.Solver(0) as s:
    s.upload(ws, capi.make_options(**fixed))
    s.solve()
    buf = (C.c_ulonglong * 16)()
    lib.mvo_hip_debug_stamps(buf, 1)
    s.solve()
    b = (C.c_ulonglong * 8)()
    lib.mvo_hip_debug_stamps_lm4(b)
    n = max(1, b[0])
    print("k_lm4,", W, "windows: turns", b[0], " cycles per turn (4 windows evaluated + one step):",
          {names[i]: b[i] // n for i in range(1, 6)}, "total", (sum(b[1:6]) - b[3]) // n, " per launch: prologue", b[6], "loop exit", b[3], "epilogue", b[7])

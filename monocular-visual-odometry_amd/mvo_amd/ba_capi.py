"""ctypes binding of the bundle-adjustment entry points of libmvo_hip.so (include/mvo_hip_ba.h).
There is no CPU fallback: without a HIP device BaSolver() raises."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

EXPORTED_SYMBOLS = [
    "mvo_hip_ba_create", "mvo_hip_ba_destroy", "mvo_hip_ba_refine", "mvo_hip_ba_upload", "mvo_hip_ba_solve",
    "mvo_hip_ba_download", "mvo_hip_ba_eval", "mvo_hip_ba_reduced", "mvo_hip_ba_set_trace",
    "mvo_hip_ba_get_trace", "mvo_hip_ba_set_profiling", "mvo_hip_ba_get_timing", "mvo_hip_ba_partial_begin",
    "mvo_hip_ba_partial_linearize", "mvo_hip_ba_partial_solve", "mvo_hip_ba_partial_decide",
    "mvo_hip_ba_partial_exchange",
]

MAX_FREE_FRAMES = 28
LOSS_NONE, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("loss", C.c_int32), ("loss_scale", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_radius", C.c_double)]


class Observations(C.Structure):
    _fields_ = [("n_landmarks", C.c_uint32), ("reserved", C.c_uint32), ("landmark_offsets", C.c_void_p),
                ("frame", C.c_void_p), ("rays", C.c_void_p)]


class Summary(C.Structure):
    _fields_ = [("termination", C.c_int32), ("iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class Iteration(C.Structure):
    _fields_ = [("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
                ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("upload_seconds", C.c_double), ("solve_seconds", C.c_double), ("device_seconds", C.c_double),
                ("schur_seconds", C.c_double), ("schur_launches", C.c_int32), ("reserved", C.c_int32),
                ("observations", C.c_uint64), ("schur_bytes", C.c_uint64),
                ("schur_flops", C.c_uint64)]


def make_options(max_num_iterations=10, loss=LOSS_NONE, loss_scale=0.01, function_tolerance=1e-6,
                 gradient_tolerance=1e-10, parameter_tolerance=1e-8, initial_radius=1e4):
    return Options(max_num_iterations, loss, loss_scale, function_tolerance, gradient_tolerance,
                   parameter_tolerance, initial_radius)


_READY = False


def load():
    global _READY
    L = capi.load()
    if not _READY:
        L.mvo_hip_ba_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mvo_hip_ba_destroy.argtypes = [C.c_void_p]
        L.mvo_hip_ba_destroy.restype = None
        L.mvo_hip_ba_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.POINTER(Observations), C.POINTER(Options)]
        L.mvo_hip_ba_refine.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.POINTER(Observations), C.POINTER(Options), C.POINTER(Summary)]
        L.mvo_hip_ba_solve.argtypes = [C.c_void_p]
        L.mvo_hip_ba_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Summary)]
        L.mvo_hip_ba_eval.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.mvo_hip_ba_reduced.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mvo_hip_ba_partial_begin.argtypes = [C.c_void_p]
        L.mvo_hip_ba_partial_linearize.argtypes = [C.c_void_p]
        L.mvo_hip_ba_partial_solve.argtypes = [C.c_void_p]
        L.mvo_hip_ba_partial_decide.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
        L.mvo_hip_ba_partial_exchange.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int]
        L.mvo_hip_ba_set_trace.argtypes = [C.c_void_p, C.c_int32]
        L.mvo_hip_ba_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.mvo_hip_ba_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.mvo_hip_ba_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        _READY = True
    return L


def _check(rc, what):
    if rc != capi.OK:
        raise capi.MvoHipError(rc, what)


class BaSolver:
    """One mvo_hip_ba_handle (one GPU, one stream)."""

    def __init__(self, device=0, profiling=True):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.mvo_hip_ba_create(int(device), C.byref(self._h))
        if rc != capi.OK:
            self._h = None
            raise capi.MvoHipError(rc, "mvo_hip_ba_create")
        self._lib.mvo_hip_ba_set_profiling(self._h, 1 if profiling else 0)
        self._keep = None
        self._shape = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mvo_hip_ba_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _obs(self, sc):
        offs = np.ascontiguousarray(sc.lm_offsets, dtype=np.uint64)
        fr = np.ascontiguousarray(sc.obs_frame, dtype=np.uint32)
        rays = np.ascontiguousarray(sc.obs_ray, dtype=np.float64)
        o = Observations(len(offs) - 1, 0, offs.ctypes.data, fr.ctypes.data, rays.ctypes.data)
        return o, (offs, fr, rays)

    def upload(self, sc, opt):
        poses = np.ascontiguousarray(sc.poses, dtype=np.float64)
        X = np.ascontiguousarray(sc.X, dtype=np.float64)
        o, keep = self._obs(sc)
        _check(self._lib.mvo_hip_ba_upload(self._h, poses.ctypes.data, sc.M, sc.N, X.ctypes.data, C.byref(o),
                                           C.byref(opt)), "mvo_hip_ba_upload")
        self._shape = (sc.N, sc.M, len(X), int(sc.lm_offsets[-1]), poses.copy())

    def solve(self):
        _check(self._lib.mvo_hip_ba_solve(self._h), "mvo_hip_ba_solve")

    def download(self):
        N, M, L, _, poses0 = self._shape
        poses = poses0.copy()
        X = np.zeros((L, 3))
        s = Summary()
        _check(self._lib.mvo_hip_ba_download(self._h, poses.ctypes.data, X.ctypes.data, C.byref(s)),
               "mvo_hip_ba_download")
        return poses, X, s

    def refine(self, sc, opt):
        poses = np.array(sc.poses, dtype=np.float64, order="C")
        X = np.array(sc.X, dtype=np.float64, order="C")
        o, keep = self._obs(sc)
        s = Summary()
        _check(self._lib.mvo_hip_ba_refine(self._h, poses.ctypes.data, sc.M, sc.N, X.ctypes.data, C.byref(o),
                                           C.byref(opt), C.byref(s)), "mvo_hip_ba_refine")
        self._shape = (sc.N, sc.M, len(X), int(sc.lm_offsets[-1]), np.array(sc.poses, dtype=np.float64))
        return poses, X, s

    def evaluate(self):
        n_obs = self._shape[3]
        r = np.zeros((n_obs, 2))
        cost = C.c_double()
        _check(self._lib.mvo_hip_ba_eval(self._h, r.ctypes.data, C.byref(cost)), "mvo_hip_ba_eval")
        return r, cost.value

    def reduced(self, radius):
        n = 6 * self._shape[1]
        S, rhs, gc = np.zeros((n, n)), np.zeros(n), np.zeros(n)
        gm, cost = C.c_double(), C.c_double()
        _check(self._lib.mvo_hip_ba_reduced(self._h, float(radius), S.ctypes.data, rhs.ctypes.data, gc.ctypes.data,
                                            C.byref(gm), C.byref(cost)), "mvo_hip_ba_reduced")
        return S, rhs, gc, gm.value, cost.value

    def solve_partitioned(self, allreduce_sum, allreduce_max, buf):
        """Landmark-partitioned solve: `buf` is a float64 device tensor (torch) of at least
        18 M (M + 1) + 18 M + 1 elements; allreduce_sum / allreduce_max reduce a tensor in place over
        the ranks (e.g. torch.distributed.all_reduce on the RCCL backend) and return once it is done."""
        M = self._shape[1]
        n0 = 18 * M * (M + 1) + 18 * M + 1
        assert buf.numel() >= n0 and buf.is_contiguous()
        ptr = buf.data_ptr()

        def xchg(which, n, op):
            _check(self._lib.mvo_hip_ba_partial_exchange(self._h, which, ptr, n, 0), "mvo_hip_ba_partial_exchange")
            op(buf[:n])
            _check(self._lib.mvo_hip_ba_partial_exchange(self._h, which, ptr, n, 1), "mvo_hip_ba_partial_exchange")

        done = C.c_int32(0)
        _check(self._lib.mvo_hip_ba_partial_begin(self._h), "mvo_hip_ba_partial_begin")
        xchg(2, 4, allreduce_sum)
        _check(self._lib.mvo_hip_ba_partial_decide(self._h, 0, C.byref(done)), "mvo_hip_ba_partial_decide")
        while not done.value:
            _check(self._lib.mvo_hip_ba_partial_linearize(self._h), "mvo_hip_ba_partial_linearize")
            xchg(0, n0, allreduce_sum)
            xchg(1, 1, allreduce_max)
            _check(self._lib.mvo_hip_ba_partial_solve(self._h), "mvo_hip_ba_partial_solve")
            xchg(2, 4, allreduce_sum)
            _check(self._lib.mvo_hip_ba_partial_decide(self._h, 1, C.byref(done)), "mvo_hip_ba_partial_decide")

    def set_trace(self, capacity):
        _check(self._lib.mvo_hip_ba_set_trace(self._h, int(capacity)), "mvo_hip_ba_set_trace")

    def trace(self, capacity=256):
        rows = (Iteration * capacity)()
        n = C.c_int32()
        _check(self._lib.mvo_hip_ba_get_trace(self._h, rows, capacity, C.byref(n)), "mvo_hip_ba_get_trace")
        return [rows[i] for i in range(min(n.value, capacity))]

    def timing(self):
        t = Timing()
        _check(self._lib.mvo_hip_ba_get_timing(self._h, C.byref(t)), "mvo_hip_ba_get_timing")
        return {f: getattr(t, f) for f, _ in Timing._fields_}

"""ctypes binding of the closed-form scale stage of libmvo_hip.so (include/mvo_hip_scaler.h), the
GPU stand-in for the arithmetic of the reference's Scaler::stage (mvo/src/pipeline/Scaler.cpp:21-68).
No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

EXPORTED_SYMBOLS = ["mvo_hip_scaler_create", "mvo_hip_scaler_destroy", "mvo_hip_scaler_scale",
                    "mvo_hip_scaler_upload", "mvo_hip_scaler_run", "mvo_hip_scaler_last_timing",
                    "mvo_hip_scaler_tracks_timing"]


class Input(C.Structure):
    _fields_ = [("R0", C.c_double * 9), ("R1", C.c_double * 9), ("R2", C.c_double * 9), ("u1", C.c_double * 3),
                ("n1", C.c_double), ("lowest_length", C.c_double), ("highest_length", C.c_double)]


class Result(C.Structure):
    _fields_ = [("scale", C.c_double), ("median", C.c_double), ("n_tracks", C.c_uint64), ("clamped", C.c_int32),
                ("reserved", C.c_int32)]


def make_input(R0, R1, R2, u1, n1, lowest_length=0.25, highest_length=2.0):
    i = Input()
    for name, a in (("R0", R0), ("R1", R1), ("R2", R2)):
        getattr(i, name)[:] = list(np.asarray(a, dtype=np.float64).reshape(9))
    i.u1[:] = list(np.asarray(u1, dtype=np.float64).reshape(3))
    i.n1, i.lowest_length, i.highest_length = float(n1), float(lowest_length), float(highest_length)
    return i


_READY = False


def load():
    global _READY
    L = capi.load()
    if not _READY:
        L.mvo_hip_scaler_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mvo_hip_scaler_destroy.argtypes = [C.c_void_p]
        L.mvo_hip_scaler_destroy.restype = None
        L.mvo_hip_scaler_scale.argtypes = [C.c_void_p, C.POINTER(Input), C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.POINTER(Result)]
        L.mvo_hip_scaler_upload.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mvo_hip_scaler_run.argtypes = [C.c_void_p, C.POINTER(Input), C.c_void_p, C.POINTER(Result)]
        L.mvo_hip_scaler_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mvo_hip_scaler_tracks_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        _READY = True
    return L


class Scaler:
    def __init__(self, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.mvo_hip_scaler_create(int(device), C.byref(self._h))
        if rc != capi.OK:
            self._h = None
            raise capi.MvoHipError(rc, "mvo_hip_scaler_create")
        self._n = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mvo_hip_scaler_destroy(self._h)
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

    @staticmethod
    def _arr(v):
        return np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 3)

    def scale(self, inp, v0, v1, v2, per_track=True):
        v0, v1, v2 = self._arr(v0), self._arr(v1), self._arr(v2)
        n = len(v0)
        out = np.zeros(n) if per_track else None
        res = Result()
        rc = self._lib.mvo_hip_scaler_scale(self._h, C.byref(inp), n, v0.ctypes.data, v1.ctypes.data, v2.ctypes.data,
                                            out.ctypes.data if per_track else None, C.byref(res))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_scaler_scale")
        return res, out

    def upload(self, v0, v1, v2):
        v0, v1, v2 = self._arr(v0), self._arr(v1), self._arr(v2)
        self._n = len(v0)
        rc = self._lib.mvo_hip_scaler_upload(self._h, self._n, v0.ctypes.data, v1.ctypes.data, v2.ctypes.data)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_scaler_upload")

    def run(self, inp, per_track=False):
        out = np.zeros(self._n) if per_track else None
        res = Result()
        rc = self._lib.mvo_hip_scaler_run(self._h, C.byref(inp), out.ctypes.data if per_track else None, C.byref(res))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_scaler_run")
        return res, out

    def last_timing(self):
        s, b = C.c_double(), C.c_uint64()
        self._lib.mvo_hip_scaler_last_timing(self._h, C.byref(s), C.byref(b))
        return s.value, b.value

    def tracks_timing(self):
        """(seconds, algorithmic bytes) of k_scaler_tracks in the last run."""
        s, b = C.c_double(), C.c_uint64()
        self._lib.mvo_hip_scaler_tracks_timing(self._h, C.byref(s), C.byref(b))
        return s.value, b.value


def synthetic_tracks(n, seed=0, noise_sigma=1e-3):
    """A 3-keyframe window as Scaler::stage sees it: rays (x, y, 1) of n landmarks in the newest
    (v0), kept (v1) and previous (v2) frame, rotations, the kept frame's baseline u1 and length n1.
    Rotations and rays are paired consistently here (R_k belongs to v_k); which rotation the
    reference's stage actually passes for which rays is the caller's business (see the header)."""
    from .scene import _small_rotation

    rng = np.random.Generator(np.random.MT19937(seed))
    R = [_small_rotation(rng, np.deg2rad(2.0)) for _ in range(3)]  # camera-to-world of frames 0 (newest), 1, 2
    step = rng.uniform(0.4, 1.6, 2)
    d = np.stack([0.35 + rng.uniform(0, 0.3, 2), rng.normal(0, 0.1, 2), np.ones(2)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = np.zeros((3, 3))
    pos[1] = pos[0] + step[0] * d[0]
    pos[2] = pos[1] + step[1] * d[1]
    z = rng.uniform(4.0, 40.0, n)
    cam = np.stack([z * np.tan(np.deg2rad(rng.uniform(-30, 30, n))), z * np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, n))), z], 1)
    X = pos[2] + cam @ R[2].T
    v = []
    for i in range(3):
        q = (X - pos[i]) @ R[i]
        q = q / q[:, 2:3] + np.concatenate([rng.normal(0, noise_sigma, (n, 2)), np.zeros((n, 1))], axis=1)
        v.append(np.ascontiguousarray(q))
    return dict(R0=R[0], R1=R[1], R2=R[2], u1=d[1], n1=step[1], v0=v[0], v1=v[1], v2=v[2], true_n0=step[0])

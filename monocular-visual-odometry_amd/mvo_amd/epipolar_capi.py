"""ctypes binding of the RANSAC baseline-direction stage of libmvo_hip.so
(include/mvo_hip_epipolar.h <- EpipolarGeometry::estimateBaseLine, EpipolarGeometry.cpp:34-176).
No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

EXPORTED_SYMBOLS = ["mvo_hip_epipolar_create", "mvo_hip_epipolar_destroy", "mvo_hip_epipolar_estimate",
                    "mvo_hip_epipolar_score", "mvo_hip_epipolar_last_timing"]


class Result(C.Structure):
    _fields_ = [("b", C.c_double * 3), ("best_probability", C.c_double), ("n_inliers", C.c_uint32),
                ("iterations", C.c_uint32), ("planned_iterations", C.c_uint32), ("best_hypothesis", C.c_uint32),
                ("exhausted", C.c_int32), ("reserved", C.c_int32)]


_READY = False


def load():
    global _READY
    L = capi.load()
    if not _READY:
        L.mvo_hip_epipolar_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mvo_hip_epipolar_destroy.argtypes = [C.c_void_p]
        L.mvo_hip_epipolar_destroy.restype = None
        L.mvo_hip_epipolar_estimate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.POINTER(Result)]
        L.mvo_hip_epipolar_score.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                             C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mvo_hip_epipolar_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        _READY = True
    return L


class Epipolar:
    def __init__(self, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.mvo_hip_epipolar_create(int(device), C.byref(self._h))
        if rc != capi.OK:
            self._h = None
            raise capi.MvoHipError(rc, "mvo_hip_epipolar_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mvo_hip_epipolar_destroy(self._h)
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
    def _prep(mhi, mt, samples):
        mhi = np.ascontiguousarray(mhi, dtype=np.float64).reshape(-1, 3)
        mt = np.ascontiguousarray(mt, dtype=np.float64).reshape(-1, 3)
        samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, 3)
        return mhi, mt, samples

    def estimate(self, mhi, mt, samples, ps=0.99, threshold=np.cos(np.deg2rad(4.0))):
        mhi, mt, samples = self._prep(mhi, mt, samples)
        n = len(mhi)
        inl = np.zeros(max(n, 1), dtype=np.uint32)
        res = Result()
        rc = self._lib.mvo_hip_epipolar_estimate(self._h, n, mhi.ctypes.data, mt.ctypes.data, len(samples),
                                                 samples.ctypes.data, float(ps), float(threshold), inl.ctypes.data,
                                                 C.byref(res))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_epipolar_estimate")
        return res, inl[:res.n_inliers].copy()

    def score(self, mhi, mt, samples, threshold=np.cos(np.deg2rad(4.0))):
        mhi, mt, samples = self._prep(mhi, mt, samples)
        K = len(samples)
        hb, hi, hp = np.zeros((K, 3)), np.zeros(K, dtype=np.uint32), np.zeros(K)
        rc = self._lib.mvo_hip_epipolar_score(self._h, len(mhi), mhi.ctypes.data, mt.ctypes.data, K,
                                              samples.ctypes.data, float(threshold), hb.ctypes.data, hi.ctypes.data,
                                              hp.ctypes.data)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_epipolar_score")
        return hb, hi, hp

    def last_timing(self):
        s, e = C.c_double(), C.c_uint64()
        self._lib.mvo_hip_epipolar_last_timing(self._h, C.byref(s), C.byref(e))
        return s.value, e.value


def synthetic_correspondences(n, seed=0, noise_sigma=1e-3, outlier_fraction=0.2, n_samples=1024):
    """What BaselineEstimator::stage hands to estimateBaseLine: rays (x, y, 1) of n tracked features in
    the previous frame (mhi) and in the new frame after un-rotation (mt), plus the index draws
    (three per hypothesis, with replacement -- distinct within a hypothesis here, so that no
    hypothesis hinges on the sign of a rounding-level determinant)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    d = np.array([0.35 + rng.uniform(0, 0.3), rng.normal(0, 0.1), 1.0])
    d /= np.linalg.norm(d)
    step = rng.uniform(0.4, 1.6)
    z = rng.uniform(4.0, 40.0, n)
    X = np.stack([z * np.tan(np.deg2rad(rng.uniform(-30, 30, n))), z * np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, n))), z], 1)
    mhi = X / X[:, 2:3]
    Y = X - step * d
    mt = Y / Y[:, 2:3]
    mt[:, :2] += rng.normal(0, noise_sigma, (n, 2))
    bad = rng.random(n) < outlier_fraction
    nb = int(bad.sum())
    mt[bad, 0] = np.tan(np.deg2rad(rng.uniform(-30, 30, nb)))
    mt[bad, 1] = np.tan(np.deg2rad(rng.uniform(-22.5, 22.5, nb)))
    samples = np.stack([rng.permutation(n)[:3] if n >= 3 else np.zeros(3, dtype=np.int64) for _ in range(n_samples)])
    return dict(mhi=np.ascontiguousarray(mhi), mt=np.ascontiguousarray(mt), samples=samples.astype(np.uint32),
                true_direction=d, outliers=bad)

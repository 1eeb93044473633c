"""ctypes wrapper of the EpipolarGeometry restatement in oracle/libmvo_oracle.so.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmvo_oracle.so"))
        _LIB.mvo_epipolar_oracle_pinlier.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def baseline(first, second, idx):
    first, second = np.ascontiguousarray(first, dtype=np.float64), np.ascontiguousarray(second, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    b = np.zeros(3)
    lib().mvo_epipolar_oracle_baseline(_p(first), _p(second), _p(idx), C.c_uint32(len(idx)), _p(b))
    return b


def estimate(mhi, mt, samples, ps, threshold):
    mhi, mt = np.ascontiguousarray(mhi, dtype=np.float64), np.ascontiguousarray(mt, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, 3)
    N, K = len(mhi), len(samples)
    b, inl = np.zeros(3), np.zeros(max(N, 1), dtype=np.uint32)
    hb, hi, hp = np.zeros((K, 3)), np.zeros(K, dtype=np.uint32), np.zeros(K)
    n_inl, its, planned, best = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    bp = C.c_double()
    ex = lib().mvo_epipolar_oracle_estimate(C.c_uint32(N), _p(mhi), _p(mt), C.c_uint32(K), _p(samples), C.c_double(ps),
                                            C.c_double(threshold), _p(b), _p(inl), C.byref(n_inl), C.byref(its),
                                            C.byref(planned), C.byref(best), C.byref(bp), _p(hb), _p(hi), _p(hp))
    return dict(b=b, inliers=inl[:n_inl.value].copy(), iterations=its.value, planned=planned.value,
                best_hypothesis=best.value, best_probability=bp.value, exhausted=ex,
                hyp_b=hb[:its.value], hyp_inliers=hi[:its.value], hyp_prob=hp[:its.value])


def score(mhi, mt, samples, threshold):
    mhi, mt = np.ascontiguousarray(mhi, dtype=np.float64), np.ascontiguousarray(mt, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, 3)
    K = len(samples)
    hb, hi, hp = np.zeros((K, 3)), np.zeros(K, dtype=np.uint32), np.zeros(K)
    lib().mvo_epipolar_oracle_score(C.c_uint32(len(mhi)), _p(mhi), _p(mt), C.c_uint32(K), _p(samples),
                                    C.c_double(threshold), _p(hb), _p(hi), _p(hp))
    return hb, hi, hp

"""ctypes access to oracle/libmvo_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (libmvo_hip + the mvo_amd host layer) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("num_threads", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("use_loss_function", C.c_int32),
        ("fix_length", C.c_int32),
        ("lowest_length", C.c_double),
        ("highest_length", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32),
        ("reason", C.c_int32),
        ("iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("reserved", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
        ("step_is_valid", C.c_int32),
        ("step_is_successful", C.c_int32),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmvo_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.mvo_oracle_reverse_scale.restype = C.c_double
        L.mvo_oracle_reverse_scale.argtypes = [C.c_double] * 3
        L.mvo_oracle_scale.restype = C.c_double
        L.mvo_oracle_scale.argtypes = [C.c_double] * 3
        L.mvo_oracle_scale_derivative.restype = C.c_double
        L.mvo_oracle_scale_derivative.argtypes = [C.c_double] * 3
        L.mvo_oracle_set_threads.argtypes = [C.c_int32]
        L.mvo_oracle_set_threads.restype = None
        L.mvo_oracle_get_threads.restype = C.c_int32
        L.mvo_oracle_num_pairs.restype = C.c_int32
        L.mvo_oracle_num_pairs.argtypes = [C.c_uint32, C.c_uint32]
        _LIB = L
    return _LIB


def make_options(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                 parameter_tolerance=1e-8, use_loss_function=False, fix_length=False,
                 lowest_length=0.25, highest_length=2.0, num_threads=4):
    """Defaults are mvo/cfg/mvo.cfg:39-53 with Refiner.cpp:72-74's 10^-x conversion."""
    return Options(max_num_iterations, num_threads, function_tolerance, gradient_tolerance,
                   parameter_tolerance, int(use_loss_function), int(fix_length), lowest_length,
                   highest_length)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _prep(w):
    frames = np.ascontiguousarray(w.frames, dtype=np.float64).copy()
    offs = np.ascontiguousarray(w.pair_offsets, dtype=np.uint64)
    m0 = np.ascontiguousarray(w.m0, dtype=np.float64)
    m1 = np.ascontiguousarray(w.m1, dtype=np.float64)
    return frames, offs, m0, m1


def set_threads(n):
    """Threads of refine()/evaluate(); 1 (default) is the serial parity path, more is for timing only."""
    lib().mvo_oracle_set_threads(int(n))


def get_threads():
    return lib().mvo_oracle_get_threads()


def reverse_scale(length, max_len=2.0, min_len=0.25):
    return lib().mvo_oracle_reverse_scale(length, max_len, min_len)


def scale(t, max_len=2.0, min_len=0.25):
    return lib().mvo_oracle_scale(t, max_len, min_len)


def scale_derivative(t, max_len=2.0, min_len=0.25):
    return lib().mvo_oracle_scale_derivative(t, max_len, min_len)


def plus(x, delta):
    x = np.ascontiguousarray(x, dtype=np.float64)
    d = np.ascontiguousarray(delta, dtype=np.float64)
    out = np.empty(3)
    lib().mvo_oracle_plus(_p(x, C.c_double), _p(d, C.c_double), _p(out, C.c_double))
    return out


def evaluate(w, opt=None, autodiff=False):
    """-> (residuals[n], jacobian[n, 3M], cost) at the initial point, raw (no loss)."""
    opt = opt or make_options()
    frames, offs, m0, m1 = _prep(w)
    n = int(offs[-1])
    res = np.zeros(n)
    jac = np.zeros((n, 3 * w.M))
    cost = C.c_double(0.0)
    fn = lib().mvo_oracle_eval_autodiff if autodiff else lib().mvo_oracle_eval
    rc = fn(_p(frames, C.c_double), C.c_uint32(w.M), C.c_uint32(w.N), C.byref(opt),
            _p(offs, C.c_uint64), _p(m0, C.c_double), _p(m1, C.c_double), _p(res, C.c_double),
            _p(jac, C.c_double), C.byref(cost))
    if rc != 0:
        raise RuntimeError(f"oracle eval failed rc={rc}")
    return res, jac, cost.value


def evaluate_at(w, point, opt=None):
    """-> (residuals[n], jacobian[n, 3M], cost), raw, at point[M, 4] = (vec, t) per refined frame."""
    opt = opt or make_options()
    frames, offs, m0, m1 = _prep(w)
    pt = np.ascontiguousarray(point, dtype=np.float64).reshape(w.M, 4)
    n = int(offs[-1])
    res = np.zeros(n)
    jac = np.zeros((n, 3 * w.M))
    cost = C.c_double(0.0)
    rc = lib().mvo_oracle_eval_at(_p(frames, C.c_double), C.c_uint32(w.M), C.c_uint32(w.N), C.byref(opt),
                                  _p(offs, C.c_uint64), _p(m0, C.c_double), _p(m1, C.c_double),
                                  _p(pt, C.c_double), _p(res, C.c_double), _p(jac, C.c_double), C.byref(cost))
    if rc != 0:
        raise RuntimeError(f"oracle eval_at failed rc={rc}")
    return res, jac, cost.value


def pair_moments(w):
    frames, offs, m0, m1 = _prep(w)
    out = np.zeros((len(offs) - 1, 6))
    rc = lib().mvo_oracle_pair_moments(_p(frames, C.c_double), C.c_uint32(w.M), C.c_uint32(w.N),
                                       _p(offs, C.c_uint64), _p(m0, C.c_double),
                                       _p(m1, C.c_double), _p(out, C.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle moments failed rc={rc}")
    return out


def refine(w, opt=None, trace_capacity=256, point=None):
    """-> (frames_out[N,13], Summary, trace list of dict).  point[M, 4] = (vec, t): explicit start."""
    opt = opt or make_options()
    frames, offs, m0, m1 = _prep(w)
    summ = Summary()
    tr = (Iteration * trace_capacity)()
    tl = C.c_int32(0)
    pt = None if point is None else np.ascontiguousarray(point, dtype=np.float64).reshape(w.M, 4)
    rc = lib().mvo_oracle_refine_from(_p(frames, C.c_double), C.c_uint32(w.M), C.c_uint32(w.N),
                                      C.byref(opt), _p(offs, C.c_uint64), _p(m0, C.c_double),
                                      _p(m1, C.c_double), None if pt is None else _p(pt, C.c_double),
                                      C.byref(summ), tr, C.c_int32(trace_capacity), C.byref(tl))
    if rc != 0:
        raise RuntimeError(f"oracle refine failed rc={rc}")
    trace = [
        {f[0]: getattr(tr[i], f[0]) for f in Iteration._fields_}
        for i in range(min(tl.value, trace_capacity))
    ]
    return frames, summ, trace

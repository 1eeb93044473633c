"""ctypes wrapper of oracle/libmvo_oracle.so's bundle-adjustment self-oracle (mvo_ba_oracle.h).
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference counterpart)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("loss", C.c_int32), ("loss_scale", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_radius", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination", C.c_int32), ("iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class Iteration(C.Structure):
    _fields_ = [("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
                ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32)]


def make_options(max_num_iterations=10, loss=0, loss_scale=0.01, function_tolerance=1e-6,
                 gradient_tolerance=1e-10, parameter_tolerance=1e-8, initial_radius=1e4):
    return Options(max_num_iterations, loss, loss_scale, function_tolerance, gradient_tolerance,
                   parameter_tolerance, initial_radius)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.path.join(_HERE, "libmvo_oracle.so"))
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _args(sc, poses, X):
    return (C.c_uint32(sc.N), ), poses, X


def residuals(sc, opt, poses=None, X=None):
    poses = np.ascontiguousarray(sc.poses if poses is None else poses, dtype=np.float64)
    X = np.ascontiguousarray(sc.X if X is None else X, dtype=np.float64)
    r = np.zeros(2 * sc.n_observations)
    cost = C.c_double()
    lib().mvo_ba_oracle_residuals(C.c_uint32(sc.N), _dp(poses), C.c_uint32(sc.n_landmarks), _dp(X),
                                  sc.lm_offsets.ctypes.data_as(C.c_void_p), sc.obs_frame.ctypes.data_as(C.c_void_p),
                                  _dp(sc.obs_ray), C.byref(opt), _dp(r), C.byref(cost))
    return r.reshape(-1, 2), cost.value


def reduced(sc, opt, radius, poses=None, X=None):
    poses = np.ascontiguousarray(sc.poses if poses is None else poses, dtype=np.float64)
    X = np.ascontiguousarray(sc.X if X is None else X, dtype=np.float64)
    n = 6 * sc.M
    S = np.zeros((n, n))
    rhs = np.zeros(n)
    g = np.zeros(n + 3 * sc.n_landmarks)
    cost = C.c_double()
    lib().mvo_ba_oracle_reduced(C.c_uint32(sc.N), C.c_uint32(sc.M), _dp(poses), C.c_uint32(sc.n_landmarks), _dp(X),
                                sc.lm_offsets.ctypes.data_as(C.c_void_p), sc.obs_frame.ctypes.data_as(C.c_void_p),
                                _dp(sc.obs_ray), C.byref(opt), C.c_double(radius), _dp(S), _dp(rhs), _dp(g),
                                C.byref(cost))
    return S, rhs, g, cost.value


def obs_jacobian(pose, X, ray):
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    ray = np.ascontiguousarray(ray, dtype=np.float64)
    r, jc, jx = np.zeros(2), np.zeros(12), np.zeros(6)
    lib().mvo_ba_oracle_obs_jacobian(_dp(pose), _dp(X), _dp(ray), _dp(r), _dp(jc), _dp(jx))
    return r, jc.reshape(2, 6), jx.reshape(2, 3)


def pose_plus(pose, delta):
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    delta = np.ascontiguousarray(delta, dtype=np.float64)
    out = np.zeros(12)
    lib().mvo_ba_oracle_pose_plus(_dp(pose), _dp(delta), _dp(out))
    return out


def refine(sc, opt, trace_capacity=256):
    poses = np.array(sc.poses, dtype=np.float64, order="C")
    X = np.array(sc.X, dtype=np.float64, order="C")
    s = Summary()
    rows = (Iteration * trace_capacity)()
    n = C.c_int32()
    lib().mvo_ba_oracle_refine(C.c_uint32(sc.N), C.c_uint32(sc.M), _dp(poses), C.c_uint32(sc.n_landmarks), _dp(X),
                               sc.lm_offsets.ctypes.data_as(C.c_void_p), sc.obs_frame.ctypes.data_as(C.c_void_p),
                               _dp(sc.obs_ray), C.byref(opt), C.byref(s), rows, C.c_int32(trace_capacity),
                               C.byref(n))
    return poses, X, s, [rows[i] for i in range(min(n.value, trace_capacity))]

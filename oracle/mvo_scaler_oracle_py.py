"""ctypes wrapper of the Scaler restatement in oracle/libmvo_oracle.so.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmvo_oracle.so"))
        _LIB.mvo_scaler_oracle_run.restype = C.c_uint64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run(R0, R1, R2, u1, n1, lowest, highest, v0, v1, v2):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (R0, R1, R2, u1, v0, v1, v2)]
    n = len(arrs[4])
    scales = np.zeros(n)
    med, sc, cl = C.c_double(), C.c_double(), C.c_int32()
    got = lib().mvo_scaler_oracle_run(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), C.c_double(n1),
                                      C.c_double(lowest), C.c_double(highest), C.c_uint64(n), _p(arrs[4]),
                                      _p(arrs[5]), _p(arrs[6]), _p(scales), C.byref(med), C.byref(sc), C.byref(cl))
    return scales, med.value, sc.value, cl.value, got

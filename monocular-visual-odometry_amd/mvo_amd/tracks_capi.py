"""ctypes binding of include/mvo_hip_tracks.h: device-resident feature-track tables and the
on-device gather of the refinement's residual blocks (the host-side chain walk of
/root/reference/mvo/src/algorithms/IterativeRefinement.cpp:116-165 and
sliding_window/Frame.hpp:258-302).  No CPU fallback: everything goes through libmvo_hip.so."""
import ctypes as C

import numpy as np

from . import capi, scaler_capi

EXPORTED_SYMBOLS = [
    "mvo_hip_tracks_create", "mvo_hip_tracks_destroy", "mvo_hip_tracks_push", "mvo_hip_tracks_append",
    "mvo_hip_tracks_reset",
    "mvo_hip_tracks_refine", "mvo_hip_tracks_gather", "mvo_hip_tracks_scale", "mvo_hip_tracks_set_profiling",
    "mvo_hip_tracks_get_timing", "mvo_hip_tracks_pair_moments",
]


class Timing(C.Structure):
    _fields_ = [("count_seconds", C.c_double), ("fill_seconds", C.c_double),
                ("residual_blocks", C.c_uint64), ("fill_bytes", C.c_uint64)]

_BOUND = False


def load():
    global _BOUND
    L = capi.load()
    if not _BOUND:
        L.mvo_hip_tracks_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mvo_hip_tracks_destroy.argtypes = [C.c_void_p]
        L.mvo_hip_tracks_destroy.restype = None
        L.mvo_hip_tracks_push.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.mvo_hip_tracks_append.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.mvo_hip_tracks_reset.argtypes = [C.c_void_p]
        L.mvo_hip_tracks_refine.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.POINTER(capi.Options), C.POINTER(capi.Summary)]
        L.mvo_hip_tracks_gather.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_uint64]
        L.mvo_hip_tracks_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(scaler_capi.Input),
                                           C.c_void_p, C.POINTER(scaler_capi.Result)]
        L.mvo_hip_tracks_pair_moments.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.mvo_hip_tracks_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.mvo_hip_tracks_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        _BOUND = True
    return L


class TrackTable:
    """One mvo_hip_tracks_handle bound to a capi.Solver (its GPU, its stream)."""

    def __init__(self, solver, max_frames, max_features):
        self._lib = load()
        self._solver = solver  # keeps the solver alive: the table uses its stream
        self._max_features = int(max_features)
        self._h = C.c_void_p()
        rc = self._lib.mvo_hip_tracks_create(solver._h, int(max_frames), int(max_features), C.byref(self._h))
        if rc != capi.OK:
            self._h = None
            raise capi.MvoHipError(rc, "mvo_hip_tracks_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mvo_hip_tracks_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def push(self, rays, pre_feature):
        """Newest keyframe: rays [n,3] (any norm), pre_feature [n] index into the previous push or < 0."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        pre = np.ascontiguousarray(pre_feature, dtype=np.int32).reshape(-1)
        if len(pre) != len(rays):
            raise ValueError("one predecessor index per ray")
        rc = self._lib.mvo_hip_tracks_push(self._h, len(rays), rays.ctypes.data, pre.ctypes.data)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_push")

    def append(self, rays):
        """Features without predecessor appended to the newest keyframe (Frame::mergeFrame)."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        rc = self._lib.mvo_hip_tracks_append(self._h, len(rays), rays.ctypes.data)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_append")

    def reset(self):
        rc = self._lib.mvo_hip_tracks_reset(self._h)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_reset")

    def gather(self, M, N):
        """-> (pair_offsets[P+1] uint64, m0[n,3], m1[n,3]) in the reference's pair order."""
        P = capi.num_pairs(M, N)
        offs = np.zeros(P + 1, dtype=np.uint64)
        rc = self._lib.mvo_hip_tracks_gather(self._h, M, N, offs.ctypes.data, None, None, 0)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_gather")
        n = int(offs[P])
        m0 = np.zeros((n, 3))
        m1 = np.zeros((n, 3))
        if n:
            rc = self._lib.mvo_hip_tracks_gather(self._h, M, N, offs.ctypes.data, m0.ctypes.data, m1.ctypes.data, n)
            if rc != capi.OK:
                raise capi.MvoHipError(rc, "mvo_hip_tracks_gather")
        return offs, m0, m1

    def pair_moments(self, frames, M):
        """S_p of every frame pair summed straight from the tables (the fused gather-and-fold) -> [P, 6]."""
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        P = capi.num_pairs(M, len(frames))
        out = np.zeros((P, 6))
        rc = self._lib.mvo_hip_tracks_pair_moments(self._h, frames.ctypes.data, int(M), len(frames), out.ctypes.data)
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_pair_moments")
        return out

    def refine(self, frames, M, options=None):
        """IterativeRefinement::refine on the newest len(frames) keyframes -> (frames_out, Summary)."""
        opt = options or capi.make_options()
        frames = np.ascontiguousarray(frames, dtype=np.float64).copy()
        summ = capi.Summary()
        rc = self._lib.mvo_hip_tracks_refine(self._h, frames.ctypes.data, int(M), len(frames), C.byref(opt),
                                             C.byref(summ))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_refine")
        return frames, summ

    def scale(self, scaler, inp, age=0, per_track=False):
        """Scaler::stage on the three keyframes age, age + 1, age + 2 pushes ago, gathered on the device
        -> (scaler_capi.Result, per-track scales or None).  `scaler`: a scaler_capi.Scaler on the same GPU."""
        scaler_capi.load()
        res = scaler_capi.Result()
        out = None
        if per_track:
            # the number of three-frame tracks is not known up front: room for every feature of the frame
            out = np.full(int(self._max_features), np.nan)
        rc = self._lib.mvo_hip_tracks_scale(self._h, scaler._h, int(age), C.byref(inp),
                                            out.ctypes.data if out is not None else None, C.byref(res))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_scale")
        return res, (out[: res.n_tracks] if out is not None else None)

    def set_profiling(self, enabled):
        rc = self._lib.mvo_hip_tracks_set_profiling(self._h, int(bool(enabled)))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_set_profiling")

    def timing(self):
        t = Timing()
        rc = self._lib.mvo_hip_tracks_get_timing(self._h, C.byref(t))
        if rc != capi.OK:
            raise capi.MvoHipError(rc, "mvo_hip_tracks_get_timing")
        return {k: getattr(t, k) for k, _ in Timing._fields_}

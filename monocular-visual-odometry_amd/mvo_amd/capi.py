"""ctypes binding of libmvo_hip's C ABI (include/mvo_hip.h).

This is plumbing for tests and bench.py; the product is the shared library.  There is no CPU
fallback anywhere in this module: if libmvo_hip.so is missing, or no HIP device is present,
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MVO_HIP_LIB_PATH: development only (A/B runs of two builds of the same library)
LIB_PATH = os.environ.get("MVO_HIP_LIB_PATH") or os.path.join(_PKG_ROOT, "libmvo_hip.so")

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_UNSUPPORTED = 2
ERR_OUT_OF_MEMORY = 3
ERR_DEVICE = 4
ERR_NO_DEVICE = 5
ERR_STATE = 6

ASSEMBLY_AUTO = 0
ASSEMBLY_PASSES = 1
ASSEMBLY_TILED = 2
ASSEMBLY_ONE_PER_WAVE = 4
ASSEMBLY_FOUR_PER_WAVE = 8

EXPORTED_SYMBOLS = [
    "mvo_hip_create", "mvo_hip_destroy", "mvo_hip_refine", "mvo_hip_refine_batch", "mvo_hip_upload",
    "mvo_hip_solve", "mvo_hip_download", "mvo_hip_eval", "mvo_hip_pair_moments", "mvo_hip_set_trace",
    "mvo_hip_set_profiling",
    "mvo_hip_get_trace", "mvo_hip_get_timing", "mvo_hip_reverse_scale", "mvo_hip_scale",
    "mvo_hip_num_pairs", "mvo_hip_status_string", "mvo_hip_version", "mvo_hip_partial_begin",
    "mvo_hip_partial_pass", "mvo_hip_partial_advance", "mvo_hip_partial_sums", "mvo_hip_partial_sums_count", "mvo_hip_debug_poison_lds",
    "mvo_hip_debug_check_guards", "mvo_hip_refine_noalloc", "mvo_hip_eval_at", "mvo_hip_debug_set_point",
    "mvo_hip_debug_pair_sums", "mvo_hip_staging", "mvo_hip_refine_staged", "mvo_hip_comm_unique_id", "mvo_hip_comm_init", "mvo_hip_comm_destroy", "mvo_hip_comm_abort", "mvo_hip_solve_partitioned",
]


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
        ("assembly_mode", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Observations(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_uint32),
        ("reserved", C.c_uint32),
        ("pair_offsets", C.POINTER(C.c_uint64)),
        ("m0", C.POINTER(C.c_double)),
        ("m1", C.POINTER(C.c_double)),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32),
        ("reason", C.c_int32),
        ("iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("clamped_scales", C.c_int32),
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


class Timing(C.Structure):
    _fields_ = [
        ("upload_seconds", C.c_double),
        ("solve_seconds", C.c_double),
        ("device_seconds", C.c_double),
        ("assembly_seconds", C.c_double),
        ("assembly_launches", C.c_int32),
        ("lm_launches", C.c_int32),
        ("residual_blocks", C.c_uint64),
        ("assembly_bytes", C.c_uint64),
        ("assembly_first_seconds", C.c_double),
    ]


class MvoHipError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: status {status} ({status_string(status)})")


_LIB = None


def load():
    """dlopen libmvo_hip.so (in-tree).  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C {_PKG_ROOT}` "
                "(python __graft_entry__.py does). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.mvo_hip_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]
        L.mvo_hip_destroy.argtypes = [C.c_void_p]
        L.mvo_hip_destroy.restype = None
        L.mvo_hip_refine.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Options),
                                     C.POINTER(Observations), C.POINTER(Summary)]
        L.mvo_hip_refine_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                           C.POINTER(Options), C.POINTER(Observations), C.POINTER(Summary)]
        L.mvo_hip_staging.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)),
                                      C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)),
                                      C.POINTER(C.c_uint64)]
        L.mvo_hip_refine_staged.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Options),
                                            C.POINTER(Summary)]
        L.mvo_hip_upload.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.POINTER(Options), C.POINTER(Observations)]
        L.mvo_hip_solve.argtypes = [C.c_void_p]
        L.mvo_hip_download.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Summary)]
        L.mvo_hip_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Options),
                                   C.POINTER(Observations), C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.mvo_hip_eval_at.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Options),
                                      C.POINTER(Observations), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.mvo_hip_comm_unique_id.argtypes = [C.c_void_p]
        L.mvo_hip_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.mvo_hip_comm_destroy.argtypes = [C.c_void_p]
        L.mvo_hip_comm_abort.argtypes = [C.c_void_p]
        L.mvo_hip_solve_partitioned.argtypes = [C.c_void_p]
        L.mvo_hip_debug_set_point.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.mvo_hip_pair_moments.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_hip_debug_pair_sums.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_hip_set_trace.argtypes = [C.c_void_p, C.c_int32]
        L.mvo_hip_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.mvo_hip_debug_poison_lds.argtypes = [C.c_int]
        L.mvo_hip_debug_check_guards.argtypes = []
        L.mvo_hip_debug_check_guards.restype = C.c_long
        L.mvo_hip_partial_begin.argtypes = [C.c_void_p]
        L.mvo_hip_partial_pass.argtypes = [C.c_void_p]
        L.mvo_hip_partial_advance.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.mvo_hip_partial_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.mvo_hip_partial_sums_count.argtypes = [C.c_void_p]
        L.mvo_hip_partial_sums_count.restype = C.c_uint64
        L.mvo_hip_refine_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Options),
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.mvo_hip_get_trace.argtypes = [C.c_void_p, C.POINTER(Iteration), C.c_int32, C.POINTER(C.c_int32)]
        L.mvo_hip_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.mvo_hip_reverse_scale.argtypes = [C.c_double] * 3
        L.mvo_hip_reverse_scale.restype = C.c_double
        L.mvo_hip_scale.argtypes = [C.c_double] * 3
        L.mvo_hip_scale.restype = C.c_double
        L.mvo_hip_num_pairs.argtypes = [C.c_uint32, C.c_uint32]
        L.mvo_hip_num_pairs.restype = C.c_uint32
        L.mvo_hip_status_string.argtypes = [C.c_int]
        L.mvo_hip_status_string.restype = C.c_char_p
        L.mvo_hip_version.restype = C.c_char_p
        _LIB = L
    return _LIB


def status_string(status):
    return load().mvo_hip_status_string(int(status)).decode()


def make_options(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                 parameter_tolerance=1e-8, use_loss_function=False, fix_length=False,
                 lowest_length=0.25, highest_length=2.0, num_threads=4, assembly_mode=ASSEMBLY_AUTO):
    """Defaults: mvo/cfg/mvo.cfg:39-53 after Refiner.cpp:72-74's 10^-x conversion."""
    return Options(max_num_iterations, num_threads, function_tolerance, gradient_tolerance,
                   parameter_tolerance, int(use_loss_function), int(fix_length), lowest_length,
                   highest_length, assembly_mode, 0)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _Packed:
    """Keeps the numpy buffers behind an Observations[] array alive."""

    def __init__(self, windows):
        self.keep = []
        self.arr = (Observations * len(windows))()
        for i, w in enumerate(windows):
            offs = np.ascontiguousarray(w.pair_offsets, dtype=np.uint64)
            m0 = np.ascontiguousarray(w.m0, dtype=np.float64)
            m1 = np.ascontiguousarray(w.m1, dtype=np.float64)
            self.keep += [offs, m0, m1]
            self.arr[i].n_pairs = len(offs) - 1
            self.arr[i].pair_offsets = offs.ctypes.data_as(C.POINTER(C.c_uint64))
            self.arr[i].m0 = _dp(m0)
            self.arr[i].m1 = _dp(m1)


class Solver:
    """One mvo_hip_handle (one GPU, one stream)."""

    def __init__(self, device=0, max_windows=0, max_frames=0, max_observations=0, profiling=True):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.mvo_hip_create(int(device), max_windows, max_frames, max_observations, C.byref(self._h))
        if rc != OK:
            self._h = None
            raise MvoHipError(rc, "mvo_hip_create")
        # the library's default is off (lowest latency); tests and the bench read device timings
        self.set_profiling(profiling)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mvo_hip_destroy(self._h)
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

    # -- the reference's operator: refine() ------------------------------------------------
    def refine(self, window, options=None):
        """-> (frames_out[N,13], Summary). window: mvo_amd.scene.Window-like."""
        opt = options or make_options()
        frames = np.ascontiguousarray(window.frames, dtype=np.float64).copy()
        pk = _Packed([window])
        summ = Summary()
        rc = self._lib.mvo_hip_refine(self._h, frames.ctypes.data, window.M, window.N, C.byref(opt), pk.arr,
                                      C.byref(summ))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_refine")
        return frames, summ

    # -- the same through the handle's pinned staging block (mvo_hip_staging / mvo_hip_refine_staged) --
    def stage(self, window):
        """What the adapter's gather loop does: write pair offsets and unit rays into the handle's pinned
        staging memory (asked for in two steps here, to exercise the grow-and-keep path)."""
        n, P = int(window.pair_offsets[-1]), len(window.pair_offsets) - 1
        po, p0, p1 = C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
        cap = C.c_uint64(0)
        half = n // 2
        for want, lo, hi in ((half, 0, half), (n, half, n)):
            rc = self._lib.mvo_hip_staging(self._h, P, want, C.byref(po), C.byref(p0), C.byref(p1), C.byref(cap))
            if rc != OK:
                raise MvoHipError(rc, "mvo_hip_staging")
            assert cap.value >= want
            if hi > lo:
                np.ctypeslib.as_array(p0, shape=(3 * cap.value,))[3 * lo:3 * hi] = window.m0[lo:hi].reshape(-1)
                np.ctypeslib.as_array(p1, shape=(3 * cap.value,))[3 * lo:3 * hi] = window.m1[lo:hi].reshape(-1)
        np.ctypeslib.as_array(po, shape=(P + 1,))[:] = window.pair_offsets

    def refine_staged(self, window, options=None):
        """-> (frames_out[N,13], Summary) from what stage() put into the staging block."""
        opt = options or make_options()
        frames = np.ascontiguousarray(window.frames, dtype=np.float64).copy()
        summ = Summary()
        rc = self._lib.mvo_hip_refine_staged(self._h, frames.ctypes.data, window.M, window.N, C.byref(opt), C.byref(summ))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_refine_staged")
        return frames, summ

    def refine_batch(self, windows, options=None):
        opt = options or make_options()
        W = len(windows)
        M, N = windows[0].M, windows[0].N
        frames = np.ascontiguousarray(np.concatenate([w.frames for w in windows], axis=0), dtype=np.float64)
        pk = _Packed(windows)
        summ = (Summary * W)()
        rc = self._lib.mvo_hip_refine_batch(self._h, W, frames.ctypes.data, M, N, C.byref(opt), pk.arr, summ)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_refine_batch")
        return frames.reshape(W, N, 13), list(summ)

    # -- split phase ----------------------------------------------------------------------
    def upload(self, windows, options=None):
        opt = options or make_options()
        self._W, self._M, self._N = len(windows), windows[0].M, windows[0].N
        self._pass_mode = bool(opt.use_loss_function) or bool(opt.assembly_mode & ASSEMBLY_PASSES)
        self._frames_out = None
        self._frames0 = np.ascontiguousarray(np.concatenate([w.frames for w in windows], axis=0),
                                             dtype=np.float64)
        pk = _Packed(windows)
        rc = self._lib.mvo_hip_upload(self._h, self._W, self._frames0.ctypes.data, self._M, self._N,
                                      C.byref(opt), pk.arr)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_upload")

    def solve(self):
        rc = self._lib.mvo_hip_solve(self._h)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_solve")

    def download(self, copy=True):
        """-> (frames[W,N,13], [Summary]).  copy=False returns views of buffers that the next
        download overwrites (what a tight replay loop wants)."""
        if getattr(self, "_frames_out", None) is None or self._frames_out.shape != self._frames0.shape:
            self._frames_out = self._frames0.copy()
            self._summ_out = (Summary * self._W)()
        rc = self._lib.mvo_hip_download(self._h, self._frames_out.ctypes.data, self._summ_out)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_download")
        frames = self._frames_out.copy() if copy else self._frames_out
        return frames.reshape(self._W, self._N, 13), (list(self._summ_out) if copy else self._summ_out)

    # -- observation-partitioned mode (one window over several GPUs) -------------------------
    def partial_sums_count(self):
        """Doubles the ranks exchange per all-reduce for the current upload: 8 per pair without a loss,
        the contracted class totals (352 per window) with one."""
        return int(self._lib.mvo_hip_partial_sums_count(self._h))

    def solve_partitioned(self, allreduce, sums):
        """Solve the uploaded SHARD together with the other ranks.  `sums` is a device tensor of
        partial_sums_count() float64 (anything with .data_ptr() and .numel()); `allreduce(sums)`
        must sum it in place over all ranks and return after the result is usable by other streams
        (torch.distributed.all_reduce + torch.cuda.synchronize()).  Every rank ends with the same
        refined frames; fetch them with download()."""
        lib, n = self._lib, int(sums.numel())

        def check(rc, what):
            if rc != OK:
                raise MvoHipError(rc, what)

        def exchange():
            check(lib.mvo_hip_partial_sums(self._h, sums.data_ptr(), n, 0), "mvo_hip_partial_sums(out)")
            allreduce(sums)
            check(lib.mvo_hip_partial_sums(self._h, sums.data_ptr(), n, 1), "mvo_hip_partial_sums(in)")

        check(lib.mvo_hip_partial_begin(self._h), "mvo_hip_partial_begin")
        done = C.c_int32(0)
        for _ in range(100000):
            exchange()
            check(lib.mvo_hip_partial_advance(self._h, C.byref(done)), "mvo_hip_partial_advance")
            if done.value or not self._pass_mode:
                break
            check(lib.mvo_hip_partial_pass(self._h), "mvo_hip_partial_pass")
        if not done.value:
            raise MvoHipError(ERR_DEVICE, "partitioned solve did not terminate")

    # -- the same with the collective inside the library (RCCL on the handle's stream) -------------
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        rc = self._lib.mvo_hip_comm_init(self._h, buf, int(rank), int(world))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_comm_init")

    def comm_destroy(self):
        rc = self._lib.mvo_hip_comm_destroy(self._h)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_comm_destroy")

    def comm_abort(self):
        """ncclCommAbort: the way out after a failed collective (no wait for the stream)."""
        rc = self._lib.mvo_hip_comm_abort(self._h)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_comm_abort")

    def solve_partitioned_rccl(self):
        rc = self._lib.mvo_hip_solve_partitioned(self._h)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_solve_partitioned")

    # -- debug / parity -------------------------------------------------------------------
    def evaluate(self, window, options=None, point=None):
        """-> (residuals[n], jacobian[n,3M], cost), raw (no loss), at the initial point or at
        point[M, 4] = (vec, t) per refined frame."""
        opt = options or make_options()
        pt = None if point is None else np.ascontiguousarray(point, dtype=np.float64).reshape(window.M, 4)
        frames = np.ascontiguousarray(window.frames, dtype=np.float64)
        pk = _Packed([window])
        n = int(window.pair_offsets[-1])
        res = np.zeros(n)
        jac = np.zeros((n, 3 * window.M))
        cost = C.c_double(0.0)
        rc = self._lib.mvo_hip_eval_at(self._h, frames.ctypes.data, window.M, window.N, C.byref(opt), pk.arr,
                                       None if pt is None else pt.ctypes.data, res.ctypes.data, jac.ctypes.data,
                                       C.byref(cost))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_eval_at")
        return res, jac, cost.value

    def set_point(self, window, point):
        """Test aid: start window `window` of the last upload at point[M, 4] = (vec, t)."""
        pt = np.ascontiguousarray(point, dtype=np.float64).reshape(self._M, 4)
        rc = self._lib.mvo_hip_debug_set_point(self._h, int(window), pt.ctypes.data)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_debug_set_point")

    def pair_moments(self):
        out = np.zeros((self._W * num_pairs(self._M, self._N), 6))
        rc = self._lib.mvo_hip_pair_moments(self._h, out.ctypes.data)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_pair_moments")
        return out

    def pair_sums(self):
        """The 8 per-pair sums the LM loop last read (include/mvo_hip.h: mvo_hip_debug_pair_sums)."""
        out = np.zeros((self._W * num_pairs(self._M, self._N), 8))
        rc = self._lib.mvo_hip_debug_pair_sums(self._h, out.ctypes.data)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_debug_pair_sums")
        return out

    def set_profiling(self, enabled):
        rc = self._lib.mvo_hip_set_profiling(self._h, 1 if enabled else 0)
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_set_profiling")

    def set_trace(self, capacity):
        rc = self._lib.mvo_hip_set_trace(self._h, int(capacity))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_set_trace")

    def trace(self, capacity=256):
        rows = (Iteration * capacity)()
        n = C.c_int32(0)
        rc = self._lib.mvo_hip_get_trace(self._h, rows, capacity, C.byref(n))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_get_trace")
        return [{f[0]: getattr(rows[i], f[0]) for f in Iteration._fields_} for i in range(min(n.value, capacity))]

    def timing(self):
        t = Timing()
        rc = self._lib.mvo_hip_get_timing(self._h, C.byref(t))
        if rc != OK:
            raise MvoHipError(rc, "mvo_hip_get_timing")
        return {f[0]: getattr(t, f[0]) for f in Timing._fields_}


def refine_noalloc(device, stream, frames_ptr, M, N, options, offsets_ptr, m0_ptr, m1_ptr, n_blocks, summary_ptr):
    """mvo_hip_refine_noalloc on DEVICE pointers (ints); returns the status code, waits for nothing."""
    return int(load().mvo_hip_refine_noalloc(int(device), stream, frames_ptr, M, N, C.byref(options), offsets_ptr,
                                             m0_ptr, m1_ptr, int(n_blocks), summary_ptr))


COMM_ID_BYTES = 128
CLASS_DOUBLES = 352  # MVO_HIP_CLASS_DOUBLES


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and ships the bytes to the other ranks)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().mvo_hip_comm_unique_id(buf)
    if rc != OK:
        raise MvoHipError(rc, "mvo_hip_comm_unique_id")
    return buf.raw


def num_pairs(M, N):
    return int(load().mvo_hip_num_pairs(M, N))


def reverse_scale(length, highest=2.0, lowest=0.25):
    return load().mvo_hip_reverse_scale(length, highest, lowest)


def scale(t, highest=2.0, lowest=0.25):
    return load().mvo_hip_scale(t, highest, lowest)


def poison_lds(device=0):
    """Test aid: NaN patterns into the LDS of every CU (include/mvo_hip.h: mvo_hip_debug_poison_lds)."""
    rc = load().mvo_hip_debug_poison_lds(device)
    if rc != 0:
        raise RuntimeError(f"mvo_hip_debug_poison_lds: status {rc}")


def guard_violations():
    """Test aid: overwritten allocation guards so far (-1: MVO_HIP_POISON_ALLOC is not set)."""
    return int(load().mvo_hip_debug_check_guards())

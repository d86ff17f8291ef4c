"""semantic-icp_amd -- MI355X-native semantic-ICP registration engine.

This Python module is plumbing only: a ctypes view of the C ABI declared in
include/sicp.h (implemented in csrc/ as hand-written gfx950 HIP kernels plus a
C++ host driver).  Tests and bench.py call the engine through it.  The C++ class
shims that keep the reference's API (SemanticIterativeClosestPoint,
EmIterativeClosestPoint, GICP, SemanticPointCloud) live in host/.

There is no CPU fallback: if libsicp.so cannot be built/loaded, or no HIP device
is present, every call raises.

The directory name contains a hyphen, so import it with
    importlib.import_module("semantic-icp_amd")
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))

MODE_GICP, MODE_EM, MODE_SEMANTIC = 0, 1, 2
SOURCE, TARGET = 0, 1
SE3_EXP, SE3_LOG, SE3_PLUS, SE3_MUL, SE3_INV = 0, 1, 2, 3, 4
LM_SEQUENCE, LM_SEQUENCE_ONE_LANE = 5, 6   # sicp_se3_device: the trust-region machine fed with given evaluations
LM_SEQUENCE_EVALS, LM_SEQUENCE_OUT = 24, 37

OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_READY = -1, -2, -3, -4
ERR_TOO_FEW_POINTS, ERR_BAD_LABEL, ERR_OUT_OF_MEMORY, ERR_INTERNAL = -5, -6, -7, -8
SUBMIT_FUSED_LABELS, SUBMIT_FRESH_FEATURES = 1, 2


class SicpParams(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("knn", C.c_int32),
        ("k_cov", C.c_int32),
        ("num_classes", C.c_int32),
        ("epsilon", C.c_double),
        ("gate_sq", C.c_double),
        ("cauchy_a", C.c_double),
        ("use_sqloss", C.c_int32),
        ("max_outer", C.c_int32),
        ("outer_tol", C.c_double),
        ("min_class_pts", C.c_int32),
        ("max_lm_iterations", C.c_int32),
        ("gradient_tolerance", C.c_double),
        ("function_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("max_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("quirk_bool_probability", C.c_int32),
        ("quirk_float_products", C.c_int32),
        ("nn_method", C.c_int32),
        ("profile", C.c_int32),
        ("lm_on_device", C.c_int32),
        ("lm_batch", C.c_int32),
        ("reuse_features", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class SicpStats(C.Structure):
    _fields_ = [
        ("outer_iters", C.c_int32),
        ("total_lm_iters", C.c_int32),
        ("total_evals", C.c_int32),
        ("weights_in_search", C.c_int32),
        ("total_corr", C.c_int64),
        ("total_active", C.c_int64),
        ("final_cost", C.c_double),
        ("t_cov_ms", C.c_double),
        ("t_nn_ms", C.c_double),
        ("t_weight_ms", C.c_double),
        ("t_solve_ms", C.c_double),
        ("t_total_ms", C.c_double),
        ("cov_kernel_ms", C.c_double),
        ("nn_kernel_ms", C.c_double),
        ("weight_kernel_ms", C.c_double),
        ("acc_kernel_ms", C.c_double),
        ("cov_launches", C.c_int32),
        ("nn_launches", C.c_int32),
        ("weight_launches", C.c_int32),
        ("acc_launches", C.c_int32),
        ("lockstep_slots", C.c_int32),
        ("graph_builds", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SicpStreamResult(C.Structure):
    _fields_ = [
        ("ticket", C.c_int64),
        ("status", C.c_int32),
        ("outer_iters", C.c_int32),
        ("qt", C.c_double * 7),
        ("stats", SicpStats),
    ]


class SicpError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = f"{where}: {_strerror(status)} ({status})"
        if detail:
            msg += f" -- {detail}"
        super().__init__(msg)


def _load_build_module():
    spec = importlib.util.spec_from_file_location("_sicp_build", os.path.join(_PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/ for gfx950 into libsicp.so (in-tree)."""
    return _load_build_module().build_lib(force=force, verbose=verbose)


LIB_PATH = os.environ.get("SICP_LIB") or os.path.join(_PKG, "libsicp.so")  # SICP_LIB: an experimental build (tuning aid)
_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)


def lib():
    """The loaded C-ABI library.  Builds it when missing and hipcc is available;
    raises if it can be neither found nor built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.sicp_strerror.restype = C.c_char_p
        _lib.sicp_last_error.restype = C.c_char_p
        _lib.sicp_last_error.argtypes = [C.c_void_p]
        _lib.sicp_version.restype = C.c_char_p
        _lib.sicp_stream_last_error.restype = C.c_char_p
        _lib.sicp_stream_last_error.argtypes = [C.c_void_p]
        for name, args in {
            "sicp_device_count": [C.POINTER(C.c_int)],
            "sicp_create": [C.c_int, C.POINTER(C.c_void_p)],
            "sicp_release_pool": [C.c_int],
            "sicp_set_memory_limit": [C.c_int, C.c_int64],
            "sicp_memory_reserved": [C.c_int, C.POINTER(C.c_int64)],
            "sicp_destroy": [C.c_void_p],
            "sicp_default_params": [C.c_int, C.POINTER(SicpParams)],
            "sicp_set_params": [C.c_void_p, C.POINTER(SicpParams)],
            "sicp_get_params": [C.c_void_p, C.POINTER(SicpParams)],
            "sicp_set_cloud": [C.c_void_p, C.c_int, C.c_int32, _fp, _fp, _fp, _up],
            "sicp_set_cloud_strided": [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64],
            "sicp_set_cloud_device": [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
            "sicp_share_cloud": [C.c_void_p, C.c_int, C.c_void_p, C.c_int],
            "sicp_cloud_size": [C.c_void_p, C.c_int, _ip, _ip],
            "sicp_set_confusion": [C.c_void_p, C.c_int32, _dp],
            "sicp_align": [C.c_void_p, _dp, _dp, _ip, C.POINTER(SicpStats)],
            "sicp_align_batch": [C.POINTER(C.c_void_p), C.c_int32, _dp, _dp, _ip, C.POINTER(SicpStats)],
            "sicp_accumulate_batch": [C.POINTER(C.c_void_p), C.c_int32, _dp, _dp, C.c_int32, _dp],
            "sicp_search_batch": [C.POINTER(C.c_void_p), C.c_int32, _dp, C.c_int32, C.c_int32, C.c_int32, _dp],
            "sicp_transform_source": [C.c_void_p, _dp, _fp, _fp, _fp],
            "sicp_fused_labels": [C.c_void_p, _dp, _up],
            "sicp_covariances": [C.c_void_p, C.c_int, _dp, _dp, _bp, _ip],
            "sicp_set_covariances": [C.c_void_p, C.c_int, _dp],
            "sicp_correspondences": [C.c_void_p, _dp, _ip, _fp, _dp],
            "sicp_accumulate": [C.c_void_p, _dp, _dp],
            "sicp_solve": [C.c_void_p, _dp, _dp, _ip, _ip, _dp],
            "sicp_se3_device": [C.c_void_p, C.c_int, C.c_int32, _dp, _dp],
            "sicp_get_stats": [C.c_void_p, C.POINTER(SicpStats)],
            "sicp_stream_create": [C.c_int, C.POINTER(SicpParams), C.c_int32, C.POINTER(C.c_void_p)],
            "sicp_stream_destroy": [C.c_void_p],
            "sicp_stream_set_confusion": [C.c_void_p, C.c_int32, _dp],
            "sicp_stream_add_cloud": [C.c_void_p, C.c_int32, _fp, _fp, _fp, _up, C.POINTER(C.c_int64)],
            "sicp_stream_add_cloud_strided": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)],
            "sicp_stream_release_cloud": [C.c_void_p, C.c_int64],
            "sicp_stream_submit": [C.c_void_p, C.c_int64, C.c_int64, _dp, C.POINTER(C.c_int64)],
            "sicp_stream_submit_ex": [C.c_void_p, C.c_int64, C.c_int64, _dp, C.c_uint32, C.POINTER(C.c_int64)],
            "sicp_stream_take_labels": [C.c_void_p, C.c_int64, C.c_int32, _up],
            "sicp_stream_poll": [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SicpStreamResult), C.POINTER(C.c_int32)],
            "sicp_stream_counters": [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
            "sicp_synchronize": [C.c_void_p],
        }.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
    return _lib


def _strerror(status: int) -> str:
    return lib().sicp_strerror(status).decode()


def version() -> str:
    return lib().sicp_version().decode()


def device_count() -> int:
    n = C.c_int(0)
    lib().sicp_device_count(C.byref(n))
    return n.value


def set_memory_limit(device: int, n_bytes: int) -> None:
    """Device memory the library may hold on `device` (0 = no limit); beyond it: SicpError(ERR_OUT_OF_MEMORY)."""
    st = lib().sicp_set_memory_limit(device, n_bytes)
    if st != OK:
        raise SicpError(st, "sicp_set_memory_limit")


def memory_reserved(device: int) -> int:
    n = C.c_int64(0)
    st = lib().sicp_memory_reserved(device, C.byref(n))
    if st != OK:
        raise SicpError(st, "sicp_memory_reserved")
    return n.value


def default_params(mode: int) -> SicpParams:
    p = SicpParams()
    st = lib().sicp_default_params(mode, C.byref(p))
    if st != OK:
        raise SicpError(st, "sicp_default_params")
    return p


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _points_and_labels(xyz, labels):
    """The caller's [n, >= 3] point array as float32 rows whose x, y, z are adjacent (any row stride; converted only
    when the dtype or the layout is something else), and the labels as uint32 (any stride)."""
    pts = np.asarray(xyz)
    if pts.ndim != 2 or pts.shape[1] < 3:
        raise ValueError("points must be an [n, 3] array")
    if pts.dtype != np.float32 or pts.shape[0] == 0 or pts.strides[1] != 4 or pts.strides[0] < 12:
        pts = np.ascontiguousarray(pts[:, :3], dtype=np.float32)
    lab = None
    if labels is not None:
        lab = np.asarray(labels)
        if lab.dtype != np.uint32 or lab.ndim != 1 or (lab.shape[0] > 1 and lab.strides[0] < 4):
            lab = np.ascontiguousarray(lab, dtype=np.uint32).reshape(-1)
        if lab.shape[0] != pts.shape[0]:
            raise ValueError("one label per point")
    return pts, lab


class Engine:
    """One handle = one MI355X + one stream (include/sicp.h)."""

    def __init__(self, device: int = 0, params: SicpParams | None = None):
        self._h = C.c_void_p()
        st = lib().sicp_create(device, C.byref(self._h))
        if st != OK:
            self._h = C.c_void_p()
            raise SicpError(st, "sicp_create")
        self.n = [0, 0]
        if params is not None:
            self.set_params(params)

    def _check(self, st, where):
        if st != OK:
            detail = lib().sicp_last_error(self._h).decode() if st in (ERR_HIP, ERR_INVALID_ARGUMENT, ERR_OUT_OF_MEMORY, ERR_INTERNAL) else ""
            raise SicpError(st, where, detail)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sicp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- configuration ----------------------------------------------------------
    def set_params(self, p: SicpParams):
        self._check(lib().sicp_set_params(self._h, C.byref(p)), "sicp_set_params")

    def get_params(self) -> SicpParams:
        p = SicpParams()
        self._check(lib().sicp_get_params(self._h, C.byref(p)), "sicp_get_params")
        return p

    def set_cloud(self, which: int, xyz, labels=None):
        pts, lab = _points_and_labels(xyz, labels)   # float32 [n, 3] as it lies in memory: no per-column copies
        self._check(
            lib().sicp_set_cloud_strided(self._h, which, pts.shape[0], pts.ctypes.data, pts.strides[0],
                                         None if lab is None else lab.ctypes.data, 0 if lab is None else lab.strides[0]),
            "sicp_set_cloud_strided",
        )
        self.n[which] = pts.shape[0]

    def set_source(self, xyz, labels=None):
        self.set_cloud(SOURCE, xyz, labels)

    def set_target(self, xyz, labels=None):
        self.set_cloud(TARGET, xyz, labels)

    def share_cloud(self, which: int, other: "Engine", other_which: int):
        """sicp_share_cloud: slot `which` refers to the device-resident cloud (tree, normals,
        histograms) in slot `other_which` of `other`; nothing is copied or rebuilt."""
        self._check(lib().sicp_share_cloud(self._h, which, other._h, other_which), "sicp_share_cloud")
        self.n[which] = other.n[other_which]

    def cloud_size(self, which: int):
        """(points handed over, finite points held in the device index)"""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(lib().sicp_cloud_size(self._h, which, C.byref(a), C.byref(b)), "sicp_cloud_size")
        return a.value, b.value

    def set_cloud_device(self, which: int, n: int, x_dev: int, y_dev: int, z_dev: int, label_dev: int | None = None):
        """sicp_set_cloud_device: SoA float32 / uint32 buffers resident on the handle's device (raw addresses)."""
        self._check(lib().sicp_set_cloud_device(self._h, which, n, x_dev, y_dev, z_dev, label_dev), "sicp_set_cloud_device")
        self.n[which] = n

    def set_confusion(self, cm):
        cm = np.ascontiguousarray(cm, dtype=np.float64)
        assert cm.ndim == 2 and cm.shape[0] == cm.shape[1]
        self._check(lib().sicp_set_confusion(self._h, cm.shape[0], _ptr(cm, _dp)), "sicp_set_confusion")

    # ---- hot path -----------------------------------------------------------------
    def align(self, init_qt=None, want_stats: bool = True):
        init = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64) if init_qt is None else np.ascontiguousarray(init_qt, dtype=np.float64)
        out = np.empty(7)
        it = C.c_int32(0)
        st = SicpStats()
        self._check(
            lib().sicp_align(self._h, _ptr(init, _dp), _ptr(out, _dp), C.byref(it), C.byref(st) if want_stats else None),
            "sicp_align",
        )
        return out, (st.as_dict() if want_stats else {"outer_iters": it.value})

    def transform_source(self, qt):
        qt = np.ascontiguousarray(qt, dtype=np.float64)
        n = self.n[SOURCE]
        ox, oy, oz = (np.empty(n, dtype=np.float32) for _ in range(3))
        self._check(lib().sicp_transform_source(self._h, _ptr(qt, _dp), _ptr(ox, _fp), _ptr(oy, _fp), _ptr(oz, _fp)), "sicp_transform_source")
        return np.stack([ox, oy, oz], axis=1)

    def fused_labels(self, qt):
        qt = np.ascontiguousarray(qt, dtype=np.float64)
        out = np.empty(self.n[SOURCE], dtype=np.uint32)
        self._check(lib().sicp_fused_labels(self._h, _ptr(qt, _dp), _ptr(out, _up)), "sicp_fused_labels")
        return out

    # ---- stage hooks -----------------------------------------------------------------
    def covariances(self, which: int, want_hist: bool = False, want_nn: bool = False):
        n = self.n[which]
        p = self.get_params()
        cov = np.empty((n, 3, 3))
        nrm = np.empty((n, 3))
        hist = np.empty((n, p.num_classes), dtype=np.uint8) if want_hist else None
        nn = np.empty((n, p.k_cov), dtype=np.int32) if want_nn else None
        self._check(lib().sicp_covariances(self._h, which, _ptr(cov, _dp), _ptr(nrm, _dp), _ptr(hist, _bp), _ptr(nn, _ip)), "sicp_covariances")
        return cov, nrm, hist, nn

    def set_covariances(self, which: int, cov9):
        """caller-supplied covariances (n x 3 x 3 or n x 9, the caller's point order); SicpError when one is not I - (1-eps) n n^T"""
        c = np.ascontiguousarray(np.asarray(cov9, dtype=np.float64).reshape(-1, 9))
        self._check(lib().sicp_set_covariances(self._h, which, _ptr(c, _dp)), "sicp_set_covariances")

    def correspondences(self, qt):
        qt = np.ascontiguousarray(qt, dtype=np.float64)
        n, K = self.n[SOURCE], self.get_params().knn
        idx = np.empty((n, K), dtype=np.int32)
        d2 = np.empty((n, K), dtype=np.float32)
        w = np.empty((n, K))
        self._check(lib().sicp_correspondences(self._h, _ptr(qt, _dp), _ptr(idx, _ip), _ptr(d2, _fp), _ptr(w, _dp)), "sicp_correspondences")
        return idx, d2, w

    def accumulate(self, qt):
        qt = np.ascontiguousarray(qt, dtype=np.float64)
        out = np.empty(28)
        self._check(lib().sicp_accumulate(self._h, _ptr(qt, _dp), _ptr(out, _dp)), "sicp_accumulate")
        return out

    def solve(self, init_qt):
        init = np.ascontiguousarray(init_qt, dtype=np.float64)
        out = np.empty(7)
        it, ev, fc = C.c_int32(), C.c_int32(), C.c_double()
        self._check(lib().sicp_solve(self._h, _ptr(init, _dp), _ptr(out, _dp), C.byref(it), C.byref(ev), C.byref(fc)), "sicp_solve")
        return out, dict(lm_iters=it.value, evals=ev.value, cost=fc.value)

    def se3_device(self, op: int, x):
        """sicp_se3_device: csrc/se3.hpp evaluated on the GPU (op = SE3_EXP / LOG / PLUS / MUL / INV), one row per item."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.shape[0]
        out = np.empty((n, LM_SEQUENCE_OUT if op >= LM_SEQUENCE else 6 if op == SE3_LOG else 7))
        self._check(lib().sicp_se3_device(self._h, op, n, _ptr(x, _dp), _ptr(out, _dp)), "sicp_se3_device")
        return out

    def stats(self):
        st = SicpStats()
        self._check(lib().sicp_get_stats(self._h, C.byref(st)), "sicp_get_stats")
        return st.as_dict()

    def synchronize(self):
        self._check(lib().sicp_synchronize(self._h), "sicp_synchronize")


# ---- lock-step batch over several engines (one per scan pair) ---------------------------------
def _handles(engines):
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    return arr


def align_batch(engines, init_qts=None, want_stats: bool = True):
    """sicp_align_batch: every engine registers its own pair, advanced in lock step.
    Returns [(qt, stats)] in the order of `engines`; per pair identical to Engine.align."""
    n = len(engines)
    init = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64), (n, 1)) if init_qts is None else \
        np.ascontiguousarray(init_qts, dtype=np.float64).reshape(n, 7)
    out = np.empty((n, 7))
    its = np.zeros(n, dtype=np.int32)
    sts = (SicpStats * n)()
    rc = lib().sicp_align_batch(_handles(engines), n, _ptr(init, _dp), _ptr(out, _dp), _ptr(its, _ip), sts if want_stats else None)
    if rc != 0:
        msgs = "; ".join(m for m in (lib().sicp_last_error(e._h).decode() for e in engines) if m)
        raise RuntimeError(f"sicp_align_batch failed: {_strerror(rc)} ({rc}) {msgs}")
    return [(out[p].copy(), (sts[p].as_dict() if want_stats else {"outer_iters": int(its[p])})) for p in range(n)]


def accumulate_batch(engines, qts, repeat: int = 1):
    """sicp_accumulate_batch: the 28 sums of every engine's current correspondences from one launch
    (issued `repeat` times back to back for timing).  Returns (out28 [n, 28], kernel_ms per launch)."""
    n = len(engines)
    qts = np.ascontiguousarray(qts, dtype=np.float64).reshape(n, 7)
    out = np.empty((n, 28))
    ms = C.c_double(0.0)
    rc = lib().sicp_accumulate_batch(_handles(engines), n, _ptr(qts, _dp), _ptr(out, _dp), repeat, C.byref(ms))
    if rc != 0:
        raise RuntimeError(f"sicp_accumulate_batch failed: {_strerror(rc)} ({rc})")
    return out, ms.value


def search_batch(engines, qts=None, what: int = 0, use_hint: bool = True, repeat: int = 1):
    """sicp_search_batch: the search kernels of every engine as one job launch (what = 0 the K-correspondence
    search at poses qts, 1 / 2 the k_cov self-search of the source / target cloud, 3 the search of 0 with the EM weights
    in its epilogue as an align() launches it), issued `repeat` times.
    Returns kernel milliseconds per repetition (all engines' searches together)."""
    n = len(engines)
    if qts is None:
        qts = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64), (n, 1))
    qts = np.ascontiguousarray(qts, dtype=np.float64).reshape(n, 7)
    ms = C.c_double(0.0)
    rc = lib().sicp_search_batch(_handles(engines), n, _ptr(qts, _dp), what, 1 if use_hint else 0, repeat, C.byref(ms))
    if rc != 0:
        msgs = "; ".join(m for m in (lib().sicp_last_error(e._h).decode() for e in engines) if m)
        raise RuntimeError(f"sicp_search_batch failed: {_strerror(rc)} ({rc}) {msgs}")
    return ms.value


class Stream:
    """sicp_stream_*: an open sequence of registrations on one device (continuous batching without the closed
    batch).  add_cloud() uploads and indexes a scan once; submit() queues source -> target; poll() hands back
    finished registrations as (ticket, qt, stats) in order of completion."""

    def __init__(self, device: int, params: SicpParams, max_in_flight: int = 256, confusion=None):
        self._s = C.c_void_p()
        st = lib().sicp_stream_create(device, C.byref(params), max_in_flight, C.byref(self._s))
        if st != OK:
            self._s = C.c_void_p()
            raise SicpError(st, "sicp_stream_create")
        if confusion is not None:
            cm = np.ascontiguousarray(confusion, dtype=np.float64)
            self._check(lib().sicp_stream_set_confusion(self._s, cm.shape[0], _ptr(cm, _dp)), "sicp_stream_set_confusion")

    def _check(self, st, where):
        if st != OK:
            raise SicpError(st, where, lib().sicp_stream_last_error(self._s).decode())

    def close(self):
        if getattr(self, "_s", None) and self._s.value:
            lib().sicp_stream_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_cloud(self, xyz, labels=None) -> int:
        pts, lab = _points_and_labels(xyz, labels)   # float32 [n, 3] as it lies in memory: no per-column copies
        cid = C.c_int64(0)
        self._check(lib().sicp_stream_add_cloud_strided(self._s, pts.shape[0], pts.ctypes.data, pts.strides[0],
                                                        None if lab is None else lab.ctypes.data, 0 if lab is None else lab.strides[0], C.byref(cid)),
                    "sicp_stream_add_cloud_strided")
        return cid.value

    def release_cloud(self, cloud_id: int):
        self._check(lib().sicp_stream_release_cloud(self._s, cloud_id), "sicp_stream_release_cloud")

    def submit(self, source_id: int, target_id: int, init_qt=None, fused_labels: bool = False, fresh_features: bool = False) -> int:
        """fused_labels: getFusedLabels at the final pose comes with the result (take_labels); fresh_features: the
        normals / histograms of both clouds are recomputed for this registration, like an align() of the reference"""
        init = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64) if init_qt is None else np.ascontiguousarray(init_qt, dtype=np.float64)
        t = C.c_int64(0)
        flags = (SUBMIT_FUSED_LABELS if fused_labels else 0) | (SUBMIT_FRESH_FEATURES if fresh_features else 0)
        if flags:
            self._check(lib().sicp_stream_submit_ex(self._s, source_id, target_id, _ptr(init, _dp), flags, C.byref(t)), "sicp_stream_submit_ex")
        else:
            self._check(lib().sicp_stream_submit(self._s, source_id, target_id, _ptr(init, _dp), C.byref(t)), "sicp_stream_submit")
        return t.value

    def take_labels(self, ticket: int, n_source: int):
        out = np.empty(n_source, dtype=np.uint32)
        self._check(lib().sicp_stream_take_labels(self._s, ticket, n_source, _ptr(out, _up)), "sicp_stream_take_labels")
        return out

    def poll(self, wait: int = 0, max_results: int = 1024):
        buf = (SicpStreamResult * max_results)()
        n = C.c_int32(0)
        self._check(lib().sicp_stream_poll(self._s, wait, max_results, buf, C.byref(n)), "sicp_stream_poll")
        return [(r.ticket, r.status, np.array(r.qt[:]), r.stats.as_dict()) for r in buf[: n.value]]

    def drain(self):
        """everything submitted so far, finished"""
        out = self.poll(wait=2)
        while True:
            more = self.poll(wait=0)
            if not more:
                return out
            out += more

    def counters(self):
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(lib().sicp_stream_counters(self._s, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "sicp_stream_counters")
        return dict(submitted=a.value, completed=b.value, busy_evals=c.value, slot_evals=d.value,
                    busy_fraction=(c.value / d.value) if d.value else 0.0)

"""ctypes binding of the CPU oracle (oracle/libsicp_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libsicp_oracle.so")

MODE_GICP, MODE_EM, MODE_SEMANTIC = 0, 1, 2


class OrcParams(C.Structure):
    _fields_ = [
        ("mode", C.c_int),
        ("knn", C.c_int),
        ("k_cov", C.c_int),
        ("epsilon", C.c_double),
        ("gate_sq", C.c_double),
        ("cauchy_a", C.c_double),
        ("use_sqloss", C.c_int),
        ("outer_tol", C.c_double),
        ("max_outer", C.c_int),
        ("min_class_pts", C.c_int),
        ("num_classes", C.c_int),
        ("gradient_tolerance", C.c_double),
        ("function_tolerance", C.c_double),
        ("max_lm_iterations", C.c_int),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("max_consecutive_invalid_steps", C.c_int),
        ("jacobi_scaling", C.c_int),
        ("num_threads", C.c_int),
        ("use_kdtree", C.c_int),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("outer_iters", C.c_int),
        ("total_lm_iters", C.c_int),
        ("total_evals", C.c_int),
        ("total_corr", C.c_int64),
        ("total_active", C.c_int64),
        ("final_cost", C.c_double),
        ("t_cov_s", C.c_double),
        ("t_nn_s", C.c_double),
        ("t_weight_s", C.c_double),
        ("t_solve_s", C.c_double),
        ("t_total_s", C.c_double),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "sicp_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "sicp_oracle.h")
    stale = (
        force
        or not os.path.exists(LIB_PATH)
        or (os.path.exists(src) and os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    )
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-B", "libsicp_oracle.so"], check=True, capture_output=True)
    return LIB_PATH


_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_em_prob.restype = C.c_double
        _lib.orc_gicp_probability.restype = C.c_int
        for name in ("orc_align", "orc_fused_labels", "orc_solve", "orc_solve_trace"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def default_params(mode: int) -> OrcParams:
    p = OrcParams()
    lib().orc_default_params(C.c_int(mode), C.byref(p))
    return p


# ---- SE3 --------------------------------------------------------------------
def se3_exp(a):
    a = _d(a)
    out = np.empty(7)
    lib().orc_se3_exp(_p(a, _dp), _p(out, _dp))
    return out


def se3_log(qt):
    qt = _d(qt)
    out = np.empty(6)
    lib().orc_se3_log(_p(qt, _dp), _p(out, _dp))
    return out


def se3_mul(a, b):
    a, b = _d(a), _d(b)
    out = np.empty(7)
    lib().orc_se3_mul(_p(a, _dp), _p(b, _dp), _p(out, _dp))
    return out


def se3_inv(a):
    a = _d(a)
    out = np.empty(7)
    lib().orc_se3_inv(_p(a, _dp), _p(out, _dp))
    return out


def se3_matrix(qt):
    qt = _d(qt)
    out = np.empty(16)
    lib().orc_se3_matrix(_p(qt, _dp), _p(out, _dp))
    return out.reshape(4, 4)


def se3_plus(qt, d):
    qt, d = _d(qt), _d(d)
    out = np.empty(7)
    lib().orc_se3_plus(_p(qt, _dp), _p(d, _dp), _p(out, _dp))
    return out


def se3_dx(qt):
    qt = _d(qt)
    out = np.empty(42)
    lib().orc_se3_dx_this_mul_exp_x_at_0(_p(qt, _dp), _p(out, _dp))
    return out.reshape(7, 6)


# ---- points -----------------------------------------------------------------
def transform_points(M, xyz):
    M = _d(M).reshape(16)
    x, y, z = (_f(xyz[:, i]) for i in range(3))
    n = x.shape[0]
    ox, oy, oz = (np.empty(n, dtype=np.float32) for _ in range(3))
    lib().orc_transform_points(_p(M, _dp), n, _p(x, _fp), _p(y, _fp), _p(z, _fp), _p(ox, _fp), _p(oy, _fp), _p(oz, _fp))
    return np.stack([ox, oy, oz], axis=1)


def knn(q, t, k, kdtree=False):
    qx, qy, qz = (_f(q[:, i]) for i in range(3))
    tx, ty, tz = (_f(t[:, i]) for i in range(3))
    nq, nt = qx.shape[0], tx.shape[0]
    idx = np.empty((nq, k), dtype=np.int32)
    d2 = np.empty((nq, k), dtype=np.float32)
    fn = lib().orc_knn_kdtree if kdtree else lib().orc_knn_brute
    fn(nq, _p(qx, _fp), _p(qy, _fp), _p(qz, _fp), nt, _p(tx, _fp), _p(ty, _fp), _p(tz, _fp), k, _p(idx, _ip), _p(d2, _fp))
    return idx, d2


def covariances(p, labels, k, eps, num_classes=0, kdtree=False):
    x, y, z = (_f(p[:, i]) for i in range(3))
    n = x.shape[0]
    cov = np.empty((n, 3, 3))
    nrm = np.empty((n, 3))
    hist = None
    lab = None
    if labels is not None:
        lab = _u(labels)
        hist = np.empty((n, num_classes))
    lib().orc_covariances(
        n, _p(x, _fp), _p(y, _fp), _p(z, _fp), _p(lab, _up), k, C.c_double(eps), num_classes, int(kdtree),
        _p(cov, _dp), _p(nrm, _dp), _p(hist, _dp),
    )
    return cov, nrm, hist


def sym3_svd_u(A):
    A = _d(A).reshape(9)
    U = np.empty(9)
    s = np.empty(3)
    lib().orc_sym3_svd_u(_p(A, _dp), _p(U, _dp), _p(s, _dp))
    return U.reshape(3, 3), s


# ---- cost function ------------------------------------------------------------
def gicp_evaluate(qt, ps, pt, Cs, Ct):
    qt, ps, pt, Cs, Ct = _d(qt), _d(ps), _d(pt), _d(Cs).reshape(9), _d(Ct).reshape(9)
    r = C.c_double()
    j = np.empty(7)
    lib().orc_gicp_evaluate(_p(qt, _dp), _p(ps, _dp), _p(pt, _dp), _p(Cs, _dp), _p(Ct, _dp), C.byref(r), _p(j, _dp))
    return r.value, j


def gicp_evaluate_local(qt, ps, pt, Cs, Ct):
    qt, ps, pt, Cs, Ct = _d(qt), _d(ps), _d(pt), _d(Cs).reshape(9), _d(Ct).reshape(9)
    r = C.c_double()
    j = np.empty(6)
    lib().orc_gicp_evaluate_local(_p(qt, _dp), _p(ps, _dp), _p(pt, _dp), _p(Cs, _dp), _p(Ct, _dp), C.byref(r), _p(j, _dp))
    return r.value, j


def gicp_probability(qt, ps, pt, Cs, Ct):
    qt, ps, pt, Cs, Ct = _d(qt), _d(ps), _d(pt), _d(Cs).reshape(9), _d(Ct).reshape(9)
    v = C.c_double()
    b = lib().orc_gicp_probability(_p(qt, _dp), _p(ps, _dp), _p(pt, _dp), _p(Cs, _dp), _p(Ct, _dp), C.byref(v))
    return bool(b), v.value


def loss(params: OrcParams, s, w=1.0):
    rho = np.empty(3)
    lib().orc_loss(C.byref(params), C.c_double(s), C.c_double(w), _p(rho, _dp))
    return rho


def em_prob(cm, t_dist, s_dist):
    cm, t_dist, s_dist = _d(cm), _d(t_dist), _d(s_dist)
    return lib().orc_em_prob(cm.shape[0], _p(cm.reshape(-1), _dp), _p(t_dist, _dp), _p(s_dist, _dp))


def accumulate(params, qt, src, scov, tgt, tcov, idx, w):
    qt = _d(qt)
    sx, sy, sz = (_f(src[:, i]) for i in range(3))
    tx, ty, tz = (_f(tgt[:, i]) for i in range(3))
    scov, tcov = _d(scov).reshape(-1), _d(tcov).reshape(-1)
    idx = _i(idx)
    K = idx.shape[1]
    w = None if w is None else _d(w)
    out = np.empty(28)
    lib().orc_accumulate(
        C.byref(params), _p(qt, _dp), sx.shape[0], _p(sx, _fp), _p(sy, _fp), _p(sz, _fp), _p(scov, _dp),
        _p(tx, _fp), _p(ty, _fp), _p(tz, _fp), _p(tcov, _dp), K, _p(idx, _ip), _p(w, _dp), _p(out, _dp),
    )
    return out


def solve(params, src, scov, tgt, tcov, idx, w, init_qt):
    sx, sy, sz = (_f(src[:, i]) for i in range(3))
    tx, ty, tz = (_f(tgt[:, i]) for i in range(3))
    scov, tcov = _d(scov).reshape(-1), _d(tcov).reshape(-1)
    idx = _i(idx)
    K = idx.shape[1]
    w = None if w is None else _d(w)
    init_qt = _d(init_qt)
    out = np.empty(7)
    it, ev, fc = C.c_int(), C.c_int(), C.c_double()
    st = lib().orc_solve(
        C.byref(params), sx.shape[0], _p(sx, _fp), _p(sy, _fp), _p(sz, _fp), _p(scov, _dp), _p(tx, _fp), _p(ty, _fp),
        _p(tz, _fp), _p(tcov, _dp), K, _p(idx, _ip), _p(w, _dp), _p(init_qt, _dp), _p(out, _dp), C.byref(it),
        C.byref(ev), C.byref(fc),
    )
    return out, dict(status=st, lm_iters=it.value, evals=ev.value, cost=fc.value)


def solve_trace(params, src, scov, tgt, tcov, idx, w, init_qt, max_trace=1000):
    """orc_solve with a record of every step attempt: (cost, radius, candidate cost, accepted)."""
    sx, sy, sz = (_f(src[:, i]) for i in range(3))
    tx, ty, tz = (_f(tgt[:, i]) for i in range(3))
    scov, tcov = _d(scov).reshape(-1), _d(tcov).reshape(-1)
    idx = _i(idx)
    K = idx.shape[1]
    w = None if w is None else _d(w)
    init_qt = _d(init_qt)
    out = np.empty(7)
    tc, trd, tcc = np.empty(max_trace), np.empty(max_trace), np.empty(max_trace)
    ta = np.empty(max_trace, dtype=np.int32)
    n = C.c_int(0)
    st = lib().orc_solve_trace(
        C.byref(params), sx.shape[0], _p(sx, _fp), _p(sy, _fp), _p(sz, _fp), _p(scov, _dp), _p(tx, _fp), _p(ty, _fp),
        _p(tz, _fp), _p(tcov, _dp), K, _p(idx, _ip), _p(w, _dp), _p(init_qt, _dp), _p(out, _dp), max_trace,
        _p(tc, _dp), _p(trd, _dp), _p(tcc, _dp), _p(ta, _ip), C.byref(n),
    )
    k = n.value
    return out, dict(status=st, cost=tc[:k], radius=trd[:k], cand_cost=tcc[:k], accepted=ta[:k])


def align(params, src, slabels, tgt, tlabels, cm, init_qt):
    sx, sy, sz = (_f(src[:, i]) for i in range(3))
    tx, ty, tz = (_f(tgt[:, i]) for i in range(3))
    sl = None if slabels is None else _u(slabels)
    tl = None if tlabels is None else _u(tlabels)
    cmf = None if cm is None else _d(cm).reshape(-1)
    init_qt = _d(init_qt)
    out = np.empty(7)
    st = OrcStats()
    rc = lib().orc_align(
        C.byref(params), sx.shape[0], _p(sx, _fp), _p(sy, _fp), _p(sz, _fp), _p(sl, _up), tx.shape[0], _p(tx, _fp),
        _p(ty, _fp), _p(tz, _fp), _p(tl, _up), _p(cmf, _dp), _p(init_qt, _dp), _p(out, _dp), C.byref(st),
    )
    if rc != 0:
        raise RuntimeError(f"orc_align failed: {rc}")
    return out, {k: getattr(st, k) for k, _ in OrcStats._fields_}


def fused_labels(params, src, slabels, tgt, tlabels, cm, qt):
    sx, sy, sz = (_f(src[:, i]) for i in range(3))
    tx, ty, tz = (_f(tgt[:, i]) for i in range(3))
    sl, tl = _u(slabels), _u(tlabels)
    cmf = _d(cm).reshape(-1)
    qt = _d(qt)
    out = np.empty(sx.shape[0], dtype=np.uint32)
    rc = lib().orc_fused_labels(
        C.byref(params), sx.shape[0], _p(sx, _fp), _p(sy, _fp), _p(sz, _fp), _p(sl, _up), tx.shape[0], _p(tx, _fp),
        _p(ty, _fp), _p(tz, _fp), _p(tl, _up), _p(cmf, _dp), _p(qt, _dp), _p(out, _up),
    )
    if rc != 0:
        raise RuntimeError(f"orc_fused_labels failed: {rc}")
    return out

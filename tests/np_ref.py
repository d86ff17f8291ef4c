"""Independent numpy/scipy statement of the semantic-icp hot-path math.

TEST INFRASTRUCTURE.  This is *not* the oracle (oracle/sicp_oracle.c) and not
the product: it is a second, independently written statement of the same
reference behaviour, used by tests/golden/make_golden.py to produce the golden
vectors that pin the C oracle.  It deliberately goes through different
machinery than the oracle: scipy.linalg.expm/logm for SE3 exp/log, LAPACK SVD
for the PCA, np.linalg.inv for the 3x3 inverses, vectorised float32 numpy for
the FLANN distance order, central finite differences for Jacobians and
scipy.optimize for the robust minimum.

Citations are into /root/reference (read while writing; never read at run time).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation

DBL_EPS = np.finfo(np.float64).eps


# ----------------------------------------------------------------------------
# SE3 in Sophus conventions: qt = [qx qy qz qw tx ty tz], tangent [upsilon; omega]
# ----------------------------------------------------------------------------
def hat6(a):
    u, w = a[:3], a[3:]
    X = np.zeros((4, 4))
    X[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    X[:3, 3] = u
    return X


def mat_to_qt(M):
    q = Rotation.from_matrix(M[:3, :3]).as_quat()  # x y z w
    if q[3] < 0:
        q = -q
    return np.concatenate([q, M[:3, 3]])


def qt_to_mat(qt):
    qt = np.asarray(qt, dtype=np.float64)
    M = np.eye(4)
    M[:3, :3] = quat_to_R(qt[:4])
    M[:3, 3] = qt[4:]
    return M


def quat_to_R(q):
    """Eigen Quaternion::toRotationMatrix, formula quoted at
    gicp_cost_function.h:110-120 (no normalisation)."""
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def se3_exp_mat(a):
    return expm(hat6(np.asarray(a, dtype=np.float64)))


def se3_log_mat(M):
    L = np.real(logm(M))
    return np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])


# ----------------------------------------------------------------------------
# pcl::transformPointCloud with a double matrix (em_icp.hpp:46-50)
# ----------------------------------------------------------------------------
def transform_points(M, xyz32):
    p = xyz32.astype(np.float64)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((M[r, 0] * p[:, 0] + M[r, 1] * p[:, 1]) + M[r, 2] * p[:, 2]) + M[r, 3]
    return out.astype(np.float32)


# ----------------------------------------------------------------------------
# exact float32 kNN in FLANN L2_Simple order; ties -> lowest index
# ----------------------------------------------------------------------------
def knn_float32(q32, t32, k):
    nq = q32.shape[0]
    idx = np.empty((nq, k), dtype=np.int32)
    d2 = np.empty((nq, k), dtype=np.float32)
    tx, ty, tz = (np.ascontiguousarray(t32[:, i]) for i in range(3))
    for i in range(nq):
        dx = q32[i, 0] - tx
        dy = q32[i, 1] - ty
        dz = q32[i, 2] - tz
        d = dx * dx
        d = d + dy * dy
        d = d + dz * dz
        assert d.dtype == np.float32
        o = np.argsort(d, kind="stable")[:k]
        idx[i] = o
        d2[i] = d[o]
    return idx, d2


# ----------------------------------------------------------------------------
# covariance + label histogram (em_icp.hpp:288-341)
# ----------------------------------------------------------------------------
def covariance_from_neighbors(p32, nn, k, eps):
    pts = p32[nn]  # float32
    mean = pts.astype(np.float64).sum(axis=0) / k
    cov = np.zeros((3, 3))
    for a in range(3):
        for b in range(a + 1):
            prod = pts[:, a] * pts[:, b]  # float32 product (quirk Q2)
            assert prod.dtype == np.float32
            v = prod.astype(np.float64).sum() / k - mean[a] * mean[b]
            cov[a, b] = v
            cov[b, a] = v
    U, s, _ = np.linalg.svd(cov)
    C = np.zeros((3, 3))
    for c in range(3):
        v = eps if c == 2 else 1.0
        C += v * np.outer(U[:, c], U[:, c])
    return C, U[:, 2], s, cov


def covariances(p32, labels, k, eps, C):
    nn, _ = knn_float32(p32, p32, k)
    n = p32.shape[0]
    covs = np.empty((n, 3, 3))
    normals = np.empty((n, 3))
    gaps = np.empty(n)
    hist = np.zeros((n, C)) if labels is not None else None
    for i in range(n):
        covs[i], normals[i], s, _ = covariance_from_neighbors(p32, nn[i], k, eps)
        gaps[i] = (s[1] - s[2]) / max(s[0], 1e-300)
        if labels is not None:
            for j in nn[i]:
                hist[i, labels[j] - 1] += 1.0 / k
    return covs, normals, hist, nn, gaps


# ----------------------------------------------------------------------------
# GICPCostFunction::Evaluate, literal (gicp_cost_function.h:27-73)
# ----------------------------------------------------------------------------
def _dR_dq(q):
    """Derivative of the un-normalised quat->R map, by complex-step
    differentiation of quat_to_R (independent of the hand-written tables at
    gicp_cost_function.h:123-173)."""
    out = []
    h = 1e-30
    for i in range(4):
        qc = np.array(q, dtype=np.complex128)
        qc[i] += 1j * h
        x, y, z, w = qc
        Rc = np.array(
            [
                [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
            ]
        )
        out.append(Rc.imag / h)
    return out  # d/dx, d/dy, d/dz, d/dw


def gicp_evaluate(qt, ps, pt, Cs, Ct):
    qt = np.asarray(qt, dtype=np.float64)
    R = quat_to_R(qt[:4])
    t = qt[4:]
    M = np.linalg.inv(Ct + R @ Cs @ R.T)
    res = pt - (R @ ps + t)
    dT = M @ res
    r = float(res @ dT)
    Ta = np.linalg.inv(Ct.T + R @ Cs.T @ R.T)
    tb = M @ res
    tc = Ta @ res
    dR = -(
        np.outer(tb, ps)
        + np.outer(tc, res @ Ta @ R @ Cs.T)
        + np.outer(tb, res @ M @ R @ Cs)
        + np.outer(tc, ps)
    )
    d = _dR_dq(qt[:4])
    jac = np.empty(7)
    for i in range(4):
        jac[i] = np.trace(dR.T @ d[i])  # order x y z w == storage order
    jac[4:] = -2.0 * dT
    return r, jac


def dx_this_mul_exp_x_at_0(qt):
    """7x6 d(T*exp(x))/dx at 0 -- by central differences through expm."""
    T = qt_to_mat(qt)
    J = np.empty((7, 6))
    h = 1e-6
    for c in range(6):
        e = np.zeros(6)
        e[c] = h
        Tp = T @ se3_exp_mat(e)
        Tm = T @ se3_exp_mat(-e)
        qp, qm = _mat_to_qt_near(Tp, qt), _mat_to_qt_near(Tm, qt)
        J[:, c] = (qp - qm) / (2 * h)
    return J


def _mat_to_qt_near(M, qt_ref):
    q = Rotation.from_matrix(M[:3, :3]).as_quat()
    if np.dot(q, qt_ref[:4]) < 0:
        q = -q
    return np.concatenate([q, M[:3, 3]])


def residual_of_pose_matrix(T, ps, pt, Cs, Ct):
    R = T[:3, :3]
    res = pt - (R @ ps + T[:3, 3])
    return float(res @ np.linalg.inv(Ct + R @ Cs @ R.T) @ res)


def gicp_local_fd(qt, ps, pt, Cs, Ct, h=1e-6):
    """d r(T exp(delta)) / d delta by central differences (what the GradientChecker
    in exec/test_gradient.cc compares the analytic Jacobian against)."""
    T = qt_to_mat(qt)
    J = np.empty(6)
    for c in range(6):
        e = np.zeros(6)
        e[c] = h
        J[c] = (
            residual_of_pose_matrix(T @ se3_exp_mat(e), ps, pt, Cs, Ct)
            - residual_of_pose_matrix(T @ se3_exp_mat(-e), ps, pt, Cs, Ct)
        ) / (2 * h)
    return J


def gicp_closed_form(R, t, ps, ns, pt, nt, eps):
    """residual + local 6-vector Jacobian from normals."""
    m = R @ ns
    A = 2 * np.eye(3) - (1 - eps) * (np.outer(nt, nt) + np.outer(m, m))
    res = pt - (R @ ps + t)
    a = np.linalg.solve(A, res)
    r = float(res @ a)
    b = R.T @ a
    c = ps + b - (1 - eps) * (ns @ b) * ns
    return r, np.concatenate([-2 * b, 2 * np.cross(b, c)])


def probability(qt, ps, pt, Cs, Ct):
    """gicp_cost_function.h:75-87 (value before the bool conversion)."""
    R = quat_to_R(np.asarray(qt)[:4])
    cov = Ct + R @ Cs @ R.T
    res = pt - (R @ ps + np.asarray(qt)[4:])
    mahal = -0.5 * float(res @ np.linalg.inv(cov) @ res)
    with np.errstate(under="ignore"):
        return float(np.linalg.det(2 * np.pi * cov) ** -0.5 * np.exp(mahal))


# ----------------------------------------------------------------------------
# losses: rho0 as closed expressions; rho1/rho2 by differentiation
# ----------------------------------------------------------------------------
def rho0(mode, s, w, a):
    if mode in ("gicp", "em"):
        v = np.sqrt(s + DBL_EPS)
        out = a * a * np.log1p(v / (a * a))
        return out * (w if mode == "em" else 1.0)
    return a * a * np.log1p(s / (a * a))


def em_prob(cm, t_dist, s_dist):
    return float((t_dist @ cm) @ (s_dist @ cm))


# ----------------------------------------------------------------------------
# robustified objective and a small independent solver
# ----------------------------------------------------------------------------
def objective(T, mode, a, src, scov, tgt, tcov, pairs, w):
    """1/2 sum rho0(r^2) over (i, j) pairs at pose matrix T."""
    R, t = T[:3, :3], T[:3, 3]
    tot = 0.0
    for e, (i, j) in enumerate(pairs):
        res = tgt[j] - (R @ src[i] + t)
        r = float(res @ np.linalg.inv(tcov[j] + R @ scov[i] @ R.T) @ res)
        tot += 0.5 * rho0(mode, r * r, w[e], a)
    return tot

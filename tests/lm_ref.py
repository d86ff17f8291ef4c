"""An independently written trust-region Levenberg-Marquardt loop for the inner solve.

TEST INFRASTRUCTURE.  Purpose: the product's LM machine (csrc/lm.hpp) and the oracle's
(oracle/sicp_oracle.c: orc_solve) are two transcriptions of one understanding of Ceres; equal
iteration counts between them prove agreement with each other only.  This file is a third
statement that shares no code and no formulation with either:

  * written from the Ceres Solver documentation ("Non-linear Least Squares" -> Trust Region
    Methods -> Levenberg-Marquardt; Solver::Options defaults; LossFunction / "Theory" for the
    robustified Gauss-Newton model), not from the oracle;
  * works on the explicit n x 6 Jacobian and residual vector (the oracle and the product only ever
    see the 28 reduced sums), robustifies them row by row the way Ceres' Corrector does, and takes
    the step by a least-squares solve (numpy lstsq = LAPACK SVD) of the STACKED system
    [J; sqrt(D/mu)] dx = [-r; 0] -- the DENSE_QR formulation the reference configures
    (em_icp.hpp:165) -- instead of a Cholesky factorisation of the normal equations;
  * residuals and Jacobians come from the closed form of SURVEY.md appendix B in vectorised numpy
    (np_ref.gicp_closed_form is its scalar twin, checked against finite differences in
    tests/test_oracle.py), poses are moved with scipy's matrix exponential.

tests/test_oracle.py compares the per-step trace (cost, radius, candidate cost, accept / reject)
of this loop with the oracle's.
"""
from __future__ import annotations

import numpy as np

from np_ref import DBL_EPS, mat_to_qt, qt_to_mat, se3_exp_mat

# Solver::Options defaults the reference leaves untouched (Ceres docs, "Solver::Options")
INITIAL_RADIUS = 1e4          # initial_trust_region_radius
MAX_RADIUS = 1e16             # max_trust_region_radius
MIN_RADIUS = 1e-32            # min_trust_region_radius
MIN_RELATIVE_DECREASE = 1e-3  # min_relative_decrease
MIN_LM_DIAGONAL = 1e-6        # min_lm_diagonal
MAX_LM_DIAGONAL = 1e32        # max_lm_diagonal
MAX_INVALID = 5               # max_num_consecutive_invalid_steps
PARAMETER_TOLERANCE = 1e-8


def residuals_and_jacobian(T, src, sn, tgt, tn, pairs, eps):
    """r_k and d r_k / d delta (T exp(delta), delta = [upsilon; omega]) for every (i, j) in pairs."""
    R, t = T[:3, :3], T[:3, 3]
    i, j = pairs[:, 0], pairs[:, 1]
    ps, ns, pt, nt = src[i], sn[i], tgt[j], tn[j]
    m = ns @ R.T
    A = 2 * np.eye(3)[None] - (1 - eps) * (nt[:, :, None] * nt[:, None, :] + m[:, :, None] * m[:, None, :])
    res = pt - (ps @ R.T + t)
    a = np.linalg.solve(A, res[:, :, None])[:, :, 0]
    r = np.einsum("ni,ni->n", res, a)
    b = a @ R                                  # R^T a
    c = ps + b - (1 - eps) * np.einsum("ni,ni->n", ns, b)[:, None] * ns
    J = np.concatenate([-2 * b, 2 * np.cross(b, c)], axis=1)
    return r, J


def loss(mode, s, w, a):
    """rho(s) and rho'(s) of the reference's loss stacks (em_icp.hpp:109-117, gicp.hpp:98-104,
    semantic_icp.hpp:96), from their definitions: Cauchy(a): a^2 log(1 + s/a^2); SQLoss: sqrt(s + eps);
    Composed(f, g)(s) = f(g(s)); Scaled(f, w) = w f."""
    b = a * a
    if mode in ("gicp", "em"):
        g = np.sqrt(s + DBL_EPS)
        rho = b * np.log1p(g / b)
        drho = (1.0 / (1.0 + g / b)) * (0.5 / g)
        scale = w if mode == "em" else 1.0
        return scale * rho, scale * drho
    return b * np.log1p(s / b), 1.0 / (1.0 + s / b)


def robustified(T, mode, a, src, sn, tgt, tn, pairs, w, eps):
    """Ceres' corrected residual vector and Jacobian (rho'' <= 0 for these losses, so the correction is
    the plain sqrt(rho') scaling), and the cost 1/2 sum rho(r^2)."""
    r, J = residuals_and_jacobian(T, src, sn, tgt, tn, pairs, eps)
    rho, drho = loss(mode, r * r, w, a)
    sq = np.sqrt(drho)
    return sq * r, sq[:, None] * J, 0.5 * float(rho.sum())


def qr_step(stacked, rhs):
    """The DENSE_QR route spelled out: Householder QR of the stacked system, then back substitution.  A zero
    pivot (an exactly rank-deficient system) is a failed linear solve -- Ceres' LevenbergMarquardtStrategy
    reports LINEAR_SOLVER_FAILURE when the step is not finite, and the minimizer counts an invalid step."""
    Q, R = np.linalg.qr(stacked)
    y = Q.T @ rhs
    d = np.zeros(R.shape[0])
    for i in range(R.shape[0] - 1, -1, -1):
        if R[i, i] == 0.0:
            return None
        d[i] = (y[i] - R[i, i + 1:] @ d[i + 1:]) / R[i, i]
    return d


def solve(mode, a, src, sn, tgt, tn, pairs, w, eps, init_qt, max_iterations=400, gradient_tolerance=1e-11,
          function_tolerance=1e-11, initial_radius=INITIAL_RADIUS, min_lm_diagonal=MIN_LM_DIAGONAL, linear_solver="lstsq"):
    """Returns (qt, trace) with trace = list of (cost, radius, candidate cost, accepted) per step attempt
    (accepted: 1 = accepted, 0 = rejected or the attempt at which a tolerance ended the solve, -1 = invalid)."""
    T = qt_to_mat(init_qt)
    f, J, cost = robustified(T, mode, a, src, sn, tgt, tn, pairs, w, eps)
    # Jacobi scaling (jacobi_scaling = true): columns scaled by 1 / (1 + ||J_col||), computed once
    col_scale = 1.0 / (1.0 + np.sqrt((J * J).sum(axis=0)))
    mu, nu = initial_radius, 2.0
    x = mat_to_qt(T)
    x_norm = np.linalg.norm(x)
    trace = []
    invalid = 0
    diag = None
    it = 0
    while it < max_iterations:
        g = J.T @ f
        # gradient test in the ambient space: || x - Plus(x, -g) ||_inf
        xm = mat_to_qt_like(T @ se3_exp_mat(-g), x)
        if np.abs(x - xm).max() <= gradient_tolerance or mu <= MIN_RADIUS:
            break
        it += 1
        Js = J * col_scale[None, :]
        if diag is None:
            diag = np.clip((Js * Js).sum(axis=0), min_lm_diagonal, MAX_LM_DIAGONAL)
        # LM step: min || Js d + f ||^2 + || sqrt(diag / mu) d ||^2 as ONE stacked least-squares problem
        stacked = np.vstack([Js, np.diag(np.sqrt(diag / mu))])
        rhs = np.concatenate([-f, np.zeros(6)])
        if linear_solver == "qr":
            d = qr_step(stacked, rhs)
        else:
            d, *_ = np.linalg.lstsq(stacked, rhs, rcond=None)
        model_cost_change = 0.0
        if d is not None:
            model = Js @ d
            model_cost_change = -float(model @ (f + 0.5 * model))
        if d is None or not (model_cost_change > 0.0) or not np.isfinite(d).all():
            trace.append((cost, mu, cost, -1))
            invalid += 1
            if invalid >= MAX_INVALID:
                break
            mu *= 0.5
            continue
        invalid = 0
        delta = d * col_scale
        T_new = T @ se3_exp_mat(delta)
        x_new = mat_to_qt_like(T_new, x)
        f_new, J_new, cost_new = robustified(T_new, mode, a, src, sn, tgt, tn, pairs, w, eps)
        trace.append([cost, mu, cost_new, 0])
        if np.linalg.norm(x - x_new) <= PARAMETER_TOLERANCE * (x_norm + PARAMETER_TOLERANCE):
            break
        if abs(cost - cost_new) <= function_tolerance * cost:
            break
        rho_q = (cost - cost_new) / model_cost_change
        if rho_q > MIN_RELATIVE_DECREASE:
            trace[-1][3] = 1
            T, x, f, J, cost = T_new, x_new, f_new, J_new, cost_new
            x_norm = np.linalg.norm(x)
            mu = min(MAX_RADIUS, mu / max(1.0 / 3.0, 1.0 - (2.0 * rho_q - 1.0) ** 3))
            nu = 2.0
            diag = None
        else:
            mu = mu / nu
            nu *= 2.0
    return x, [tuple(t) for t in trace]


def mat_to_qt_like(T, ref):
    q = mat_to_qt(T)
    if np.dot(q[:4], ref[:4]) < 0:
        q[:4] = -q[:4]
    return q

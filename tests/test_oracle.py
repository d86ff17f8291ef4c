"""CPU tests: the C oracle against the committed golden vectors
(tests/golden/*.npz, produced by tests/golden/make_golden.py from the
independent numpy/scipy statement in tests/np_ref.py)."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import oracle_lib as O
import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def pose_err(qt, T):
    D = np.linalg.inv(T) @ O.se3_matrix(qt)
    return np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()), np.linalg.norm(D[:3, 3])


def test_se3_against_expm_logm():
    g = load("se3.npz")
    for i in range(g["A"].shape[0]):
        qt = O.se3_exp(g["A"][i])
        assert np.allclose(O.se3_matrix(qt), g["exp_mats"][i], atol=1e-14, rtol=0)
        assert np.allclose(O.se3_log(qt), g["A"][i], atol=1e-13, rtol=0)
        qb = O.se3_exp(g["B"][i])
        assert np.allclose(O.se3_matrix(O.se3_mul(qt, qb)), g["prod"][i], atol=1e-14, rtol=0)
        assert np.allclose(O.se3_matrix(O.se3_inv(qt)), g["inv"][i], atol=1e-14, rtol=0)
        assert np.allclose(O.se3_matrix(O.se3_plus(qt, g["B"][i])), g["prod"][i], atol=1e-14, rtol=0)
        assert np.allclose(O.se3_dx(qt), g["dx"][i], atol=2e-9, rtol=0)  # golden is a central difference
        assert abs(np.linalg.norm(O.se3_mul(qt, qb)[:4]) - 1) < 1e-15


def test_cost_function_literal_and_local():
    g = load("costfn.npz")
    n = g["qts"].shape[0]
    for k in range(n):
        a = (g["qts"][k], g["ps"][k], g["pt"][k], g["Cs"][k], g["Ct"][k])
        r, j7 = O.gicp_evaluate(*a)
        assert np.isclose(r, g["residual"][k], rtol=1e-11, atol=1e-13)
        assert np.allclose(j7, g["jac7"][k], rtol=1e-10, atol=1e-10 * np.abs(g["jac7"][k]).max())
        r2, j6 = O.gicp_evaluate_local(*a)
        assert r2 == r
        fd = g["jac6_fd"][k]
        assert np.allclose(j6, fd, rtol=0, atol=2e-7 * max(1.0, np.abs(fd).max()))
        if g["eps"][k] > 0:  # closed form of SURVEY appendix B == literal chain rule
            assert np.isclose(r, g["r_closed"][k], rtol=1e-10, atol=1e-12)
            jc = g["j_closed"][k]
            assert np.allclose(j6, jc, rtol=0, atol=1e-9 * max(1.0, np.abs(jc).max()))
        b, v = O.gicp_probability(*a)
        gv = g["prob"][k]
        if np.isnan(gv):
            assert np.isnan(v) and b  # quirk Q1: NaN converts to true
        else:
            assert np.isclose(v, gv, rtol=1e-9, atol=0) and b == (gv != 0.0)


def test_probability_underflow_gate():
    # quirk Q1: the "probability" only gates on double underflow (Mahalanobis^2 >~ 1490)
    Cs = np.eye(3) - 0.999 * np.outer([0, 0, 1], [0, 0, 1])
    I = np.array([0, 0, 0, 1, 0, 0, 0.0])
    ps = np.zeros(3)
    for dz, expect in ((1.0, True), (1.7, True), (1.75, False), (3.0, False)):
        b, v = O.gicp_probability(I, ps, np.array([0, 0, dz]), Cs, Cs)
        assert b == expect, (dz, v)


def test_knn_brute_kdtree_numpy_identical():
    g = load("knn.npz")
    q, t = g["q"], g["t"]
    for k in (1, 4, 20):
        ib, db = O.knn(q, t, k)
        ik, dk = O.knn(q, t, k, kdtree=True)
        assert np.array_equal(ib, g[f"idx{k}"]) and np.array_equal(db, g[f"d2_{k}"])
        assert np.array_equal(ik, ib) and np.array_equal(dk, db)
    # fewer targets than k: tail is (-1, inf)
    ib, db = O.knn(q[:3], t[:2], 4)
    assert (ib[:, 2:] == -1).all() and np.isinf(db[:, 2:]).all() and (ib[:, :2] >= 0).all()
    assert np.array_equal(O.transform_points(g["M"], q), g["q_transformed"])


def test_covariances_and_histograms():
    g = load("cov.npz")
    cov, nrm, hist = O.covariances(g["p"], g["labels"], int(g["k"]), float(g["eps"]), int(g["C"]))
    ok = g["gaps"] > 1e-6  # PCA direction is only defined away from degenerate spectra
    assert ok.mean() > 0.99
    assert np.allclose(cov[ok], g["cov"][ok], atol=1e-9, rtol=0)
    dots = np.abs(np.einsum("ni,ni->n", nrm, g["normals"]))
    assert (1 - dots[ok]).max() < 1e-12
    assert np.array_equal(hist, g["hist"])
    assert np.allclose(hist.sum(axis=1), 1.0, atol=1e-12)
    # C = I - (1-eps) n n^T  
    eps = float(g["eps"])
    rebuilt = np.eye(3)[None] - (1 - eps) * np.einsum("ni,nj->nij", nrm, nrm)
    assert np.allclose(cov, rebuilt, atol=1e-14, rtol=0)
    ck, _, _ = O.covariances(g["p"], g["labels"], int(g["k"]), eps, int(g["C"]), kdtree=True)
    assert np.array_equal(ck, cov)


def test_sym3_svd_orders_by_absolute_value():
    Q, _ = np.linalg.qr(np.random.default_rng(5).normal(size=(3, 3)))
    A = Q @ np.diag([2.0, 3e-4, -5e-4]) @ Q.T  # float-product noise can make cov slightly indefinite
    U, s = O.sym3_svd_u(A)
    Un, sn, _ = np.linalg.svd(A)
    assert np.allclose(s, sn, atol=1e-15)
    assert np.allclose(np.abs(np.sum(U * Un, axis=0)), 1.0, atol=1e-12)


@pytest.mark.parametrize("mode,name", [(O.MODE_GICP, "gicp"), (O.MODE_EM, "em"), (O.MODE_SEMANTIC, "semantic")])
def test_losses(mode, name):
    g = load("loss.npz")
    p = O.default_params(mode)
    for s, want in zip(g["s"], g[name]):
        rho = O.loss(p, float(s), float(g["w"]))
        # Ceres evaluates b*log(1 + s/b) (not log1p): absolute error ~ b*2^-53 near s = 0
        assert np.isclose(rho[0], want[0], rtol=1e-9, atol=4e-15), (s, rho, want)
        assert np.allclose(rho[1:], want[1:], rtol=1e-9, atol=1e-300), (s, rho, want)
        assert rho[2] <= 0  # => Ceres' Corrector always takes the simple branch


def test_em_prob():
    g = load("em.npz")
    for cm, want in ((g["cm"], g["p1"]), (g["cm2"], g["p2"])):
        got = np.array([O.em_prob(cm, g["td"][i], g["sd"][i]) for i in range(g["td"].shape[0])])
        assert np.allclose(got, want, rtol=1e-13, atol=0)


def _align(mode, g, eps=1e-3, init=None, kdtree=True):
    p = O.default_params(mode)
    p.num_classes = 4
    p.epsilon = eps
    p.use_kdtree = int(kdtree)
    I = np.array([0, 0, 0, 1, 0, 0, 0.0]) if init is None else init
    return O.align(p, g["src"], g["sl"], g["tgt"], g["tl"], g["cm"], I)


@pytest.mark.parametrize("mode,key", [(O.MODE_GICP, "gicp"), (O.MODE_EM, "em"), (O.MODE_SEMANTIC, "sem")])
def test_align_matches_independent_solver(mode, key):
    """Full align(): oracle LM (Ceres-style) vs scipy least_squares outer loop."""
    g = load("align.npz")
    qt, st = _align(mode, g)
    assert st["outer_iters"] == int(g[f"{key}_outer"])
    rot, tr = pose_err(qt, g[f"{key}_T"])
    assert rot < 1e-6 and tr < 1e-6, (rot, tr)
    rot, tr = pose_err(qt, g["T_gt"])  # and it actually registers the pair
    assert rot < 2e-3 and tr < 1e-2
    qb, sb = _align(mode, g, kdtree=False)
    assert np.array_equal(qb, qt) and sb["total_evals"] == st["total_evals"]


def test_align_em_eps1e2_nonidentity_start():
    g = load("align.npz")
    from np_ref import mat_to_qt

    qt, st = _align(O.MODE_EM, g, eps=1e-2, init=mat_to_qt(g["em2_T0"]))
    assert st["outer_iters"] == int(g["em2_outer"])
    rot, tr = pose_err(qt, g["em2_T"])
    assert rot < 1e-6 and tr < 1e-6, (rot, tr)


def test_fused_labels_shape_and_range():
    g = load("align.npz")
    p = O.default_params(O.MODE_EM)
    p.num_classes = 4
    qt, _ = _align(O.MODE_EM, g)
    lab = O.fused_labels(p, g["src"], g["sl"], g["tgt"], g["tl"], g["cm"], qt)
    assert lab.shape == g["sl"].shape and lab.min() >= 1 and lab.max() <= 4
    assert (lab == g["sl"]).mean() > 0.95


@pytest.mark.parametrize("mode", ["gicp", "em", "semantic"])
def test_lm_step_control_against_an_independent_loop(mode):
    """The oracle's trust-region loop, step attempt by step attempt, against tests/lm_ref.py: a loop
    written from the Ceres documentation on the explicit Jacobian, with a stacked least-squares
    (DENSE_QR-style) step instead of Cholesky on the normal equations and scipy's expm for the
    pose update.  Same accept / reject sequence, same radii, same costs."""
    import lm_ref

    g = load("align.npz")
    src, sl, tgt, tl, cm = g["src"], g["sl"], g["tgt"], g["tl"], g["cm"]
    C = cm.shape[0]
    omode = {"gicp": O.MODE_GICP, "em": O.MODE_EM, "semantic": O.MODE_SEMANTIC}[mode]
    p = O.default_params(omode)
    p.num_classes = C
    K = 4 if mode == "em" else 1
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    scov, sn, sh = O.covariances(src, sl if mode == "em" else None, 20, p.epsilon, C)
    tcov, tn, th = O.covariances(tgt, tl if mode == "em" else None, 20, p.epsilon, C)
    idx, d2 = O.knn(src, tgt, K)
    idx[~(d2 < np.float32(p.gate_sq))] = -1
    w = np.zeros(idx.shape)
    for i in range(len(src)):
        for c in range(K):
            j = idx[i, c]
            if j < 0:
                continue
            if mode == "em":
                b, _ = O.gicp_probability(ident, src[i].astype(np.float64), tgt[j].astype(np.float64), scov[i], tcov[j])
                w[i, c] = O.em_prob(cm, th[j], sh[i]) * float(b)
            else:
                w[i, c] = 1.0
    oq, tr = O.solve_trace(p, src, scov, tgt, tcov, idx, w, ident)
    live = idx >= 0
    pairs = np.stack([np.nonzero(live)[0], idx[live]], axis=1)
    rq, rtr = lm_ref.solve(mode, p.cauchy_a, src.astype(np.float64), sn, tgt.astype(np.float64), tn, pairs, w[live], p.epsilon, ident,
                           gradient_tolerance=p.gradient_tolerance, function_tolerance=p.function_tolerance)
    assert len(rtr) == len(tr["cost"]) and len(rtr) > 5
    ref = np.array(rtr)
    assert np.array_equal(ref[:, 3].astype(int), tr["accepted"])           # same accept / reject / invalid sequence
    assert np.allclose(ref[:, 0], tr["cost"], rtol=1e-9, atol=0)            # cost at every accepted iterate
    assert np.allclose(ref[:, 2], tr["cand_cost"], rtol=1e-9, atol=0)       # cost at every candidate
    # trust-region radius of every step (near convergence rho is a ratio of differences of nearly equal
    # costs, so the two float64 implementations drift apart in the 5th digit there)
    assert np.allclose(ref[:, 1], tr["radius"], rtol=2e-4, atol=0)
    assert np.allclose(ref[:6, 1], tr["radius"][:6], rtol=1e-9, atol=0)
    D = np.linalg.inv(O.se3_matrix(oq)) @ O.se3_matrix(rq)
    assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < 1e-9 and np.linalg.norm(D[:3, 3]) < 1e-9


def _solve_inputs(src, sl, tgt, tl, cm, mode, K, p):
    C = cm.shape[0]
    scov, sn, sh = O.covariances(src, sl if mode == "em" else None, 20, p.epsilon, C)
    tcov, tn, th = O.covariances(tgt, tl if mode == "em" else None, 20, p.epsilon, C)
    idx, d2 = O.knn(src, tgt, K)
    idx[~(d2 < np.float32(p.gate_sq))] = -1
    return scov, sn, sh, tcov, tn, th, idx


def _compare_traces(tr, rtr, oq, rq, pose_tol=1e-8, radius_tail_rtol=2e-4):
    ref = np.array(rtr)
    assert len(rtr) == len(tr["cost"])
    assert np.array_equal(ref[:, 3].astype(int), tr["accepted"])           # same accept / reject / invalid sequence
    assert np.allclose(ref[:, 0], tr["cost"], rtol=1e-9, atol=0)
    assert np.allclose(ref[:, 2], tr["cand_cost"], rtol=1e-9, atol=0)
    # (near convergence the step quality is a ratio of differences of nearly equal costs: the radii of two
    # float64 implementations drift apart there)
    assert np.allclose(ref[:6, 1], tr["radius"][:6], rtol=1e-9, atol=0)
    assert np.allclose(ref[:, 1], tr["radius"], rtol=radius_tail_rtol, atol=0)
    D = np.linalg.inv(O.se3_matrix(oq)) @ O.se3_matrix(rq)
    assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < pose_tol and np.linalg.norm(D[:3, 3]) < pose_tol


@pytest.mark.parametrize("data", ["golden", "lidar20k"])
@pytest.mark.parametrize("mode", ["gicp", "em", "semantic"])
def test_lm_rejected_steps_against_the_independent_loop(mode, data):
    """The REJECTED-step branch of the trust-region loop (radius /= nu, nu *= 2, the LM diagonal kept; then the
    recovery after an accepted step: nu = 2, new diagonal) compared step attempt by step attempt with
    tests/lm_ref.py.  Clean correspondences from a good start never reject a Gauss-Newton step, so the inner
    solve is started far from the optimum on a correspondence set full of wrong matches (what the first outer
    iteration of a badly initialised ICP hands to Ceres): runs of consecutive rejections, then progress again.
    Two data sets: the golden pair and the 20K-point LiDAR pair."""
    import lm_ref
    from np_ref import mat_to_qt

    if data == "golden":
        g = load("align.npz")
        src, sl, tgt, tl, cm = g["src"], g["sl"], g["tgt"], g["tl"], g["cm"]
        frac, start, seed = 0.9, synth.pose_matrix(60.0, (1, 2, 3), (3.0, -2.0, 1.0)), 0
    else:
        src, sl, tgt, tl, T_gt, cm = synth.lidar_pair(seed=2, n_points=20000)
        frac, start, seed = 0.3, synth.pose_matrix(25.0, (1, 2, 3), (2.0, -1.0, 0.5)), 1
    omode = {"gicp": O.MODE_GICP, "em": O.MODE_EM, "semantic": O.MODE_SEMANTIC}[mode]
    p = O.default_params(omode)
    p.num_classes = cm.shape[0]
    # On the LiDAR pair this landscape keeps the solver zig-zagging for 100+ attempts, whose accept / reject
    # decisions near the end hang on the last digits of nearly equal costs; the first 40 attempts hold a run
    # of eight rejections and the recovery after it, and both loops then stop at the iteration cap.
    cap = 400 if data == "golden" else 40
    p.max_lm_iterations = cap
    K = 4 if mode == "em" else 1
    scov, sn, sh, tcov, tn, th, idx = _solve_inputs(src, sl, tgt, tl, cm, mode, K, p)
    rng = np.random.default_rng(seed)
    wrong = rng.random(idx.shape) < frac
    idx[wrong] = rng.integers(0, len(tgt), int(wrong.sum()))
    w = np.where(idx >= 0, rng.uniform(0.2, 1.0, idx.shape) if mode == "em" else 1.0, 0.0)
    init = mat_to_qt(start)
    oq, tr = O.solve_trace(p, src, scov, tgt, tcov, idx, w, init)
    acc = tr["accepted"]
    rejected = np.nonzero(acc[:-1] == 0)[0]           # (the last attempt may be the one a tolerance stopped)
    assert len(rejected) >= 2 and (np.diff(rejected) == 1).any(), acc   # at least one RUN of rejections (nu doubles)
    assert (acc[rejected[0]:] == 1).any()                               # and accepted steps after it
    live = idx >= 0
    pairs = np.stack([np.nonzero(live)[0], idx[live]], axis=1)
    rq, rtr = lm_ref.solve(mode, p.cauchy_a, src.astype(np.float64), sn, tgt.astype(np.float64), tn, pairs, w[live], p.epsilon, init,
                           gradient_tolerance=p.gradient_tolerance, function_tolerance=p.function_tolerance, max_iterations=cap)
    _compare_traces(tr, rtr, oq, rq)


@pytest.mark.parametrize("mode", ["gicp", "semantic"])
def test_lm_invalid_steps_against_the_independent_loop(mode):
    """The INVALID-step branch (the linear solve fails: radius halved, the attempt repeated without a new
    evaluation, termination after max_consecutive_invalid_steps).  A registration problem that lives in the
    plane z = 0 -- points, normals and start pose -- has three Jacobian columns that are exactly zero
    (z translation, x and y rotation); with min_lm_diagonal = 0 nothing regularises them, Cholesky meets a
    zero pivot and the QR route a zero diagonal of R: every attempt is invalid, five in a row end the solve.
    With the default min_lm_diagonal the same problem is solved; both against tests/lm_ref.py."""
    import lm_ref
    from np_ref import mat_to_qt

    rng = np.random.default_rng(3)
    n = 400
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = 5.0 + 0.5 * np.sin(3 * ang)
    tgt = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.zeros(n)], 1).astype(np.float32)
    tn = np.stack([np.cos(ang + 0.1), np.sin(ang + 0.1), np.zeros(n)], 1)
    T = synth.pose_matrix(4.0, (0, 0, 1), (0.15, -0.1, 0.0))                  # a motion inside the plane
    src = (tgt.astype(np.float64) - T[:3, 3]) @ T[:3, :3]
    src[:, :2] += rng.normal(0, 0.02, (n, 2))                                  # in-plane noise: the minimum is not at cost 0
    src = src.astype(np.float32)
    src[:, 2] = 0.0
    sn = tn @ T[:3, :3]
    sn[:, 2] = 0.0
    sn /= np.linalg.norm(sn, axis=1, keepdims=True)
    omode = {"gicp": O.MODE_GICP, "semantic": O.MODE_SEMANTIC}[mode]
    p = O.default_params(omode)
    eye = np.eye(3)[None]
    scov = eye - (1 - p.epsilon) * sn[:, :, None] * sn[:, None, :]
    tcov = eye - (1 - p.epsilon) * tn[:, :, None] * tn[:, None, :]
    idx = np.arange(n, dtype=np.int32)[:, None]
    w = np.ones((n, 1))
    pairs = np.stack([np.arange(n), np.arange(n)], 1)
    init = mat_to_qt(synth.pose_matrix(1.0, (0, 0, 1), (0.05, 0.02, 0.0)))
    for min_diag, solver in ((0.0, "qr"), (1e-6, "qr")):
        p.min_lm_diagonal = min_diag
        oq, tr = O.solve_trace(p, src, scov, tgt, tcov, idx, w, init)
        rq, rtr = lm_ref.solve(mode, p.cauchy_a, src.astype(np.float64), sn, tgt.astype(np.float64), tn, pairs, w[:, 0], p.epsilon, init,
                               gradient_tolerance=p.gradient_tolerance, function_tolerance=p.function_tolerance,
                               min_lm_diagonal=min_diag, linear_solver=solver)
        if min_diag == 0.0:
            assert list(tr["accepted"]) == [-1] * 5 and tr["status"] == 2       # LM_INVALID_STEPS
            assert np.allclose(tr["radius"], 1e4 * 0.5 ** np.arange(5), rtol=0, atol=0)
            assert np.array_equal(oq, init)
        else:
            assert (tr["accepted"] == 1).sum() >= 3 and tr["status"] == 0
        _compare_traces(tr, rtr, oq, rq, radius_tail_rtol=1e-2)

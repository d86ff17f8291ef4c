"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs, against the committed goldens, and -- at the BASELINE metric size --
through size-independent properties.

Bars: bit-exact for index / integer work (kNN indices, float32 squared distances, label
histograms, fused labels); documented floating-point tolerances elsewhere; final pose within the
north-star tolerance (1e-4 rad / 1e-3 m) of the oracle -- in practice ~1e-9.
"""
import importlib
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import oracle_lib as O
import synth
from checks import assert_normals_match
from np_ref import mat_to_qt

pytestmark = pytest.mark.gpu

sicp = importlib.import_module("semantic-icp_amd")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IDENT = np.array([0, 0, 0, 1, 0, 0, 0.0])

# north star: final pose within 1e-4 rad / 1e-3 m of the reference solve
ROT_TOL, TRANS_TOL = 1e-4, 1e-3


def pose_delta(qa, qb):
    D = np.linalg.inv(O.se3_matrix(qa)) @ O.se3_matrix(qb)
    return np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()), np.linalg.norm(D[:3, 3])


def pose_err_to_matrix(q, T):
    D = np.linalg.inv(T) @ O.se3_matrix(q)
    return np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()), np.linalg.norm(D[:3, 3])


def make_engine(mode, C=0, cm=None, **kw):
    p = sicp.default_params(mode)
    p.num_classes = C
    for k, v in kw.items():
        setattr(p, k, v)
    e = sicp.Engine(0, p)
    if cm is not None:
        e.set_confusion(cm)
    return e, p


def oracle_params(mode, C=0, **kw):
    p = O.default_params(mode)
    p.num_classes = C
    p.use_kdtree = 1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.fixture(params=[0, 1, 2], ids=["bruteforce", "boxtree", "boxtree_per_query"])
def nn(request):
    """Both exact kNN engines (sicp_params.nn_method) must give bit-identical results."""
    return request.param


@pytest.fixture(params=[0, 1], ids=["hostlm", "devicelm"])
def lm(request):
    """Host-loop and device-resident inner solve run the same LM machine (csrc/lm.hpp)."""
    return request.param


@pytest.fixture(scope="module")
def pair1():
    return synth.config1_pair(seed=1, n_per_label=700)


@pytest.fixture(scope="module")
def lidar20k():
    return synth.lidar_pair(seed=2, n_points=20000)


# ------------------------------------------------------------------------------------------------
# correspondence search: bit-exact
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,K", [(sicp.MODE_GICP, 1), (sicp.MODE_EM, 4)])
def test_knn_bit_exact_vs_oracle(lidar20k, mode, K, nn):
    src, sl, tgt, tl, T_gt, cm = lidar20k
    e, p = make_engine(mode, 11, cm, gate_sq=1e30, nn_method=nn)
    e.set_source(src, sl)
    e.set_target(tgt, tl)
    qt = mat_to_qt(synth.pose_matrix(1.3, (0.1, 0.2, 1.0), (0.4, -0.2, 0.05)))
    idx, d2, _ = e.correspondences(qt)
    q = O.transform_points(O.se3_matrix(qt), src)
    oi, od = O.knn(q, tgt, K, kdtree=True)
    assert np.array_equal(idx, oi)
    assert np.array_equal(d2, od)
    assert (np.diff(d2, axis=1) >= 0).all()  # ascending, like FLANN's result set


def test_knn_ties_ragged_sizes_and_gate(nn):
    g = np.load(os.path.join(G, "knn.npz"))
    q, t = g["q"], g["t"]  # contains exact duplicate targets: lowest index must win
    e, p = make_engine(sicp.MODE_EM, 3, synth.confusion_matrix(3), gate_sq=4.0, nn_method=nn)
    lab_q = np.ones(len(q), dtype=np.uint32)
    lab_t = np.ones(len(t), dtype=np.uint32)
    e.set_source(q, lab_q)
    e.set_target(t, lab_t)
    idx, d2, _ = e.correspondences(IDENT)
    want_i, want_d = g["idx4"].copy(), g["d2_4"]
    want_i[~(want_d < np.float32(4.0))] = -1  # strict <
    assert np.array_equal(idx, want_i) and np.array_equal(d2, want_d)
    # GICP K=1, targets fewer than one LDS tile, queries not a multiple of the block
    e2, _ = make_engine(sicp.MODE_GICP, nn_method=nn)
    e2.set_source(q[:257])
    e2.set_target(t[:37])
    idx, d2, _ = e2.correspondences(IDENT)
    oi, od = O.knn(q[:257], t[:37], 1)
    oi[~(od < np.float32(250.0))] = -1
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)


def test_too_few_targets_and_bad_labels_are_errors():
    e, p = make_engine(sicp.MODE_EM, 3, synth.confusion_matrix(3))
    pts = np.random.default_rng(0).normal(size=(50, 3)).astype(np.float32)
    e.set_source(pts, np.ones(50, dtype=np.uint32))
    e.set_target(pts[:3], np.ones(3, dtype=np.uint32))  # K = 4 > 3 (reference: UB, em_icp.hpp:62-65)
    with pytest.raises(sicp.SicpError) as err:
        e.align()
    assert err.value.status == sicp.ERR_TOO_FEW_POINTS
    e.set_target(pts, np.zeros(50, dtype=np.uint32))  # label 0 (reference: dist(-1), UB)
    with pytest.raises(sicp.SicpError) as err:
        e.align()
    assert err.value.status == sicp.ERR_BAD_LABEL
    e3, _ = make_engine(sicp.MODE_EM, 3)  # no confusion matrix
    e3.set_source(pts, np.ones(50, dtype=np.uint32))
    e3.set_target(pts, np.ones(50, dtype=np.uint32))
    with pytest.raises(sicp.SicpError) as err:
        e3.align()
    assert err.value.status == sicp.ERR_NOT_READY


# ------------------------------------------------------------------------------------------------
# covariances / normals / histograms
# ------------------------------------------------------------------------------------------------
def test_covariances_vs_oracle_and_golden(lidar20k, nn):
    g = np.load(os.path.join(G, "cov.npz"))
    e, p = make_engine(sicp.MODE_EM, int(g["C"]), synth.confusion_matrix(int(g["C"])), nn_method=nn)
    e.set_source(g["p"], g["labels"])
    cov, nrm, hist, nbr = e.covariances(sicp.SOURCE, want_hist=True, want_nn=True)
    assert np.array_equal(nbr, g["nn"])                       # k = 20 self-kNN: bit exact
    assert np.array_equal(hist.astype(np.float64) / 20.0, g["hist"]) or np.allclose(hist / 20.0, g["hist"], atol=1e-15)
    assert np.array_equal(hist, np.rint(g["hist"] * 20).astype(np.uint8))
    ok = g["gaps"] > 1e-6
    dots = np.abs(np.einsum("ni,ni->n", nrm, g["normals"]))
    assert (1 - dots[ok]).max() < 1e-12
    assert np.allclose(cov[ok], g["cov"][ok], atol=1e-9, rtol=0)
    # bigger, LiDAR-like cloud against the oracle (float-product quirk matters at 40 m range)
    src, sl, *_ = lidar20k
    e2, _ = make_engine(sicp.MODE_EM, 11, synth.confusion_matrix(11), nn_method=nn)
    e2.set_source(src, sl)
    cov, nrm, hist, nbr = e2.covariances(sicp.SOURCE, want_hist=True, want_nn=True)
    ocov, onrm, ohist = O.covariances(src, sl, 20, 1e-3, 11, kdtree=True)
    onn, _ = O.knn(src, src, 20, kdtree=True)
    assert np.array_equal(nbr, onn)
    assert np.array_equal(hist, np.rint(ohist * 20).astype(np.uint8))
    # normals agree wherever the PCA direction is determined: every mismatch has a degenerate spectrum
    assert_normals_match(nrm, onrm, src, nbr)
    assert np.median(np.abs(cov - ocov).reshape(len(src), -1).max(axis=1)) < 1e-12


def test_small_class_divides_by_k_quirk(nn):
    # quirk Q3: fewer than k points -> still divided by k (em_icp.hpp:317,320)
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(12, 3)).astype(np.float32)
    e, _ = make_engine(sicp.MODE_GICP, nn_method=nn)
    e.set_source(pts)
    cov, nrm, _, nbr = e.covariances(sicp.SOURCE, want_nn=True)
    ocov, onrm, _ = O.covariances(pts, None, 20, 1e-3)
    assert (nbr[:, 12:] == -1).all() and (np.sort(nbr[:, :12], axis=1) == np.arange(12)).all()
    assert (1 - np.abs(np.einsum("ni,ni->n", nrm, onrm))).max() < 1e-10


# ------------------------------------------------------------------------------------------------
# weights + accumulation
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [sicp.MODE_GICP, sicp.MODE_EM, sicp.MODE_SEMANTIC])
def test_weights_and_accumulate_vs_oracle(pair1, mode, nn, lm):
    src, sl, tgt, tl, T_gt = pair1
    C = 4
    cm = synth.confusion_matrix(C)
    e, p = make_engine(mode, C, cm, nn_method=nn, lm_on_device=lm)
    e.set_source(src, sl)
    e.set_target(tgt, tl)
    qt = mat_to_qt(synth.pose_matrix(1.0, (0, 1, 0), (0.05, 0.0, -0.02)))
    idx, d2, w = e.correspondences(qt)
    op = oracle_params(mode, C)
    if mode == sicp.MODE_SEMANTIC:
        # per-label covariances and per-label search, in the caller's point order
        scov = np.zeros((len(src), 3, 3)); tcov = np.zeros((len(tgt), 3, 3))
        want = np.full((len(src), 1), -1, dtype=np.int32)
        q = O.transform_points(O.se3_matrix(qt), src)
        for l in np.unique(sl):
            si, ti = np.nonzero(sl == l)[0], np.nonzero(tl == l)[0]
            scov[si] = O.covariances(src[si], None, 20, 1e-3)[0]
            tcov[ti] = O.covariances(tgt[ti], None, 20, 1e-3)[0]
            oi, od = O.knn(q[si], tgt[ti], 1)
            want[si, 0] = np.where(od[:, 0] < np.float32(250), ti[oi[:, 0]], -1)
        assert np.array_equal(idx, want)
        ow = (idx >= 0).astype(np.float64)
    else:
        scov, _, sh = O.covariances(src, sl if mode == sicp.MODE_EM else None, 20, 1e-3, C)
        tcov, _, th = O.covariances(tgt, tl if mode == sicp.MODE_EM else None, 20, 1e-3, C)
        ow = np.zeros(idx.shape)
        for i in range(len(src)):
            for c in range(idx.shape[1]):
                j = idx[i, c]
                if j < 0:
                    continue
                if mode == sicp.MODE_EM:
                    b, _ = O.gicp_probability(qt, src[i].astype(np.float64), tgt[j].astype(np.float64), scov[i], tcov[j])
                    ow[i, c] = O.em_prob(cm, th[j], sh[i]) * float(b)
                else:
                    ow[i, c] = 1.0
    assert np.allclose(w, ow, rtol=1e-12, atol=0)
    got = e.accumulate(qt)
    ref = O.accumulate(op, qt, src, scov, tgt, tcov, idx, ow)
    scale = np.abs(ref[:21]).max()
    # closed-form Jacobian + normals vs the literal chain rule on full covariance matrices:
    # float64 throughout, different operation order
    assert np.allclose(got[:21], ref[:21], rtol=0, atol=1e-9 * scale)
    assert np.allclose(got[21:27], ref[21:27], rtol=0, atol=1e-9 * np.abs(ref[21:27]).max() + 1e-9 * scale)
    assert np.isclose(got[27], ref[27], rtol=1e-11)
    # run-to-run reproducible (no atomics in the reduction)
    assert np.array_equal(got, e.accumulate(qt))
    # inner solve from the same correspondences
    est, info = e.solve(qt)
    oest, oinfo = O.solve(op, src, scov, tgt, tcov, idx, ow, qt)
    rot, tr = pose_delta(est, oest)
    assert rot < 1e-7 and tr < 1e-7
    assert np.isclose(info["cost"], oinfo["cost"], rtol=1e-9)
    assert info["lm_iters"] == oinfo["lm_iters"]  # same trust-region path as the Ceres-style oracle


@pytest.mark.parametrize("C,bool_q", [(11, 1), (4, 1), (16, 1), (13, 0)])
def test_em_weights_from_the_search_epilogue_equal_the_weight_kernel(lidar20k, C, bool_q):
    """EM-ICP, K = 4, at most 16 classes: the packet search writes the slots' weights in its epilogue (KnnArgs::w_*) instead
    of em_weight_rows4_kernel running behind it -- the same operations in the same order (em_icp.hpp:84-89,108), so the same
    bits.  A handle with a profiling mask takes the separate kernel: both, on the same correspondences, compared bit for bit
    (odd and even class counts: padded projection rows; quirk Q1 on and off: the literal pow / exp path)."""
    src, sl, tgt, tl, T, _ = lidar20k
    rng = np.random.default_rng(C)
    sl2, tl2 = rng.integers(1, C + 1, len(sl)).astype(np.uint32), rng.integers(1, C + 1, len(tl)).astype(np.uint32)
    cm = synth.confusion_matrix(C)
    qt = mat_to_qt(T)
    got = []
    for profile in (0, 4):     # 0: the epilogue; SICP_PROFILE_WEIGHT: the weight kernel, timed
        e, p = make_engine(sicp.MODE_EM, C, cm, quirk_bool_probability=bool_q, profile=profile)
        e.set_source(src, sl2)
        e.set_target(tgt, tl2)
        st0 = e.stats()
        idx, d2, w = e.correspondences(qt)
        st1 = e.stats()
        assert (st1["weight_launches"] - st0["weight_launches"]) == (1 if profile else 0)
        got.append((idx, d2, w))
        e.close()
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert np.array_equal(got[0][2], got[1][2])          # bit for bit
    if C == 11:
        # a pair alone: only its first search -- queued beside the feature kernels, before the projections exist -- is
        # followed by the weight kernel, every later outer iteration's weights come from the search itself; a batch of more
        # than 4 pairs keeps the kernel (it hides beside the accumulate launches there).  Same poses either way.
        ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
        es = []
        for k in range(5):
            e, p = make_engine(sicp.MODE_EM, C, cm)
            e.set_source(src, sl2)
            e.set_target(tgt, tl2)
            es.append(e)
        q1, s1 = es[0].align(ident)
        assert s1["outer_iters"] >= 2 and s1["weight_launches"] == 1
        assert s1["weights_in_search"] == s1["outer_iters"] - 1   # the counters say where the weights were computed
        res = sicp.align_batch(es)
        for q, st in res:
            assert np.array_equal(q, q1) and st["weight_launches"] == st["outer_iters"] and st["weights_in_search"] == 0
        for e in es:
            e.close()
    assert (got[0][2][got[0][0] >= 0] > 0).all() and (got[0][2][got[0][0] < 0] == 0).all()


# ------------------------------------------------------------------------------------------------
# full align(): the three reference classes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,key", [(sicp.MODE_GICP, "gicp"), (sicp.MODE_EM, "em"), (sicp.MODE_SEMANTIC, "sem")])
def test_align_vs_oracle_and_golden(mode, key, nn, lm):
    g = np.load(os.path.join(G, "align.npz"))
    C = 4
    e, p = make_engine(mode, C, g["cm"], nn_method=nn, lm_on_device=lm)
    e.set_source(g["src"], g["sl"])
    e.set_target(g["tgt"], g["tl"])
    qt, st = e.align(IDENT)
    oq, ost = O.align(oracle_params(mode, C), g["src"], g["sl"], g["tgt"], g["tl"], g["cm"], IDENT)
    rot, tr = pose_delta(qt, oq)
    assert rot < ROT_TOL and tr < TRANS_TOL
    assert rot < 1e-7 and tr < 1e-7, (rot, tr)  # what is actually achieved
    assert st["outer_iters"] == ost["outer_iters"] == int(g[f"{key}_outer"])
    assert st["total_active"] == ost["total_active"] and st["total_corr"] == ost["total_corr"]
    rot, tr = pose_err_to_matrix(qt, g[f"{key}_T"])  # independent scipy solve
    assert rot < 1e-6 and tr < 1e-6
    # a12: final cloud = float(matrix) * source
    out = e.transform_source(qt)
    M = O.se3_matrix(qt).astype(np.float32)
    want = (g["src"] @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
    assert np.allclose(out, want, atol=2e-6)


def test_align_config1_all_modes(pair1):
    src, sl, tgt, tl, T_gt = pair1
    for mode in (sicp.MODE_SEMANTIC, sicp.MODE_GICP, sicp.MODE_EM):
        e, p = make_engine(mode, 4, synth.confusion_matrix(4))
        e.set_source(src, sl)
        e.set_target(tgt, tl)
        qt, st = e.align()
        oq, ost = O.align(oracle_params(mode, 4), src, sl, tgt, tl, synth.confusion_matrix(4), IDENT)
        rot, tr = pose_delta(qt, oq)
        assert rot < 1e-7 and tr < 1e-7 and st["outer_iters"] == ost["outer_iters"]
        rot, tr = pose_err_to_matrix(qt, T_gt)
        assert rot < 3e-3 and tr < 2e-2


def test_semantic_skips_small_and_missing_classes(nn):
    src, sl, tgt, tl, T_gt = synth.config1_pair(seed=5, n_per_label=450)
    # class 7: only in the source; class 9: in both but <= 400 source points
    rng = np.random.default_rng(1)
    extra_s = rng.uniform(0, 5, (500, 3)).astype(np.float32)
    small = rng.uniform(0, 5, (300, 3)).astype(np.float32)
    src2 = np.concatenate([src, extra_s, small]); sl2 = np.concatenate([sl, np.full(500, 7), np.full(300, 9)]).astype(np.uint32)
    tgt2 = np.concatenate([tgt, small + 0.5]); tl2 = np.concatenate([tl, np.full(300, 9)]).astype(np.uint32)
    e, p = make_engine(sicp.MODE_SEMANTIC, nn_method=nn)
    e.set_source(src2, sl2)
    e.set_target(tgt2, tl2)
    idx, d2, w = e.correspondences(IDENT)
    assert (idx[sl2 == 7] == -1).all() and (idx[sl2 == 9] == -1).all() and (idx[sl2 <= 4] >= 0).all()
    qt, st = e.align()
    oq, ost = O.align(oracle_params(sicp.MODE_SEMANTIC), src2, sl2, tgt2, tl2, None, IDENT)
    rot, tr = pose_delta(qt, oq)
    assert rot < 1e-7 and tr < 1e-7 and st["outer_iters"] == ost["outer_iters"]


def test_em_lidar20k_vs_oracle_and_fused_labels(lidar20k, nn):
    src, sl, tgt, tl, T_gt, cm = lidar20k
    e, p = make_engine(sicp.MODE_EM, 11, cm, nn_method=nn)
    e.set_source(src, sl)
    e.set_target(tgt, tl)
    qt, st = e.align()
    oq, ost = O.align(oracle_params(sicp.MODE_EM, 11), src, sl, tgt, tl, cm, IDENT)
    rot, tr = pose_delta(qt, oq)
    assert rot < ROT_TOL and tr < TRANS_TOL
    assert st["outer_iters"] == ost["outer_iters"]
    rot, tr = pose_err_to_matrix(qt, T_gt)
    assert rot < 5e-3 and tr < 5e-2
    lab = e.fused_labels(qt)
    olab = O.fused_labels(oracle_params(sicp.MODE_EM, 11), src, sl, tgt, tl, cm, qt)
    assert np.array_equal(lab, olab)  # same products, same summation order as em_icp.hpp:243-265: same arg-max


def test_config3_rgbd_eps1e6_13_classes_vs_oracle():
    """BASELINE config 3 settings (exec/scenenet_eval.cc:174: EmIterativeClosestPoint<13>(20, 1e-6)) on a
    strided RGB-D frame pair, EM and SemanticICP (exec/nyu_eval.cc:139), against the oracle."""
    src, sl, tgt, tl, T_gt, cm = synth.rgbd_pair(seed=3, stride=4)  # 160x120 -> 19200 points
    e, p = make_engine(sicp.MODE_EM, 13, cm, epsilon=1e-6)
    e.set_source(src, sl)
    e.set_target(tgt, tl)
    qt, st = e.align()
    oq, ost = O.align(oracle_params(sicp.MODE_EM, 13, epsilon=1e-6), src, sl, tgt, tl, cm, IDENT)
    rot, tr = pose_delta(qt, oq)
    assert rot < ROT_TOL and tr < TRANS_TOL, (rot, tr)
    assert st["outer_iters"] == ost["outer_iters"] and st["total_active"] == ost["total_active"]
    lab = e.fused_labels(qt)
    olab = O.fused_labels(oracle_params(sicp.MODE_EM, 13, epsilon=1e-6), src, sl, tgt, tl, cm, qt)
    assert np.array_equal(lab, olab)
    e2, _ = make_engine(sicp.MODE_SEMANTIC)
    e2.set_source(src, sl)
    e2.set_target(tgt, tl)
    qt2, st2 = e2.align()
    oq2, ost2 = O.align(oracle_params(sicp.MODE_SEMANTIC), src, sl, tgt, tl, None, IDENT)
    rot, tr = pose_delta(qt2, oq2)
    assert rot < ROT_TOL and tr < TRANS_TOL and st2["outer_iters"] == ost2["outer_iters"]


def test_config4_facets_20_classes_vs_oracle():
    """BASELINE config 4 settings (20 classes, full EM outer loop) on a 30K-point facets pair."""
    src, sl, tgt, tl, T_gt, cm = synth.facets_pair(seed=4, n_points=30000, n_facets=40, cube=30.0)
    e, p = make_engine(sicp.MODE_EM, 20, cm)
    e.set_source(src, sl)
    e.set_target(tgt, tl)
    qt, st = e.align()
    oq, ost = O.align(oracle_params(sicp.MODE_EM, 20), src, sl, tgt, tl, cm, IDENT)
    rot, tr = pose_delta(qt, oq)
    assert rot < ROT_TOL and tr < TRANS_TOL, (rot, tr)
    assert st["outer_iters"] == ost["outer_iters"]
    rot, tr = pose_err_to_matrix(qt, T_gt)
    assert rot < 2e-3 and tr < 2e-2


def test_quirk_flags_change_results(pair1):
    src, sl, tgt, tl, T_gt = pair1
    far = src + np.float32(35.0)  # float32 products only matter away from the origin
    e, _ = make_engine(sicp.MODE_GICP)
    e.set_source(far)
    _, n1, _, nbr = e.covariances(sicp.SOURCE, want_nn=True)
    e2, _ = make_engine(sicp.MODE_GICP, quirk_float_products=0)
    e2.set_source(far)
    _, n2, _, _ = e2.covariances(sicp.SOURCE)
    d = 1 - np.abs(np.einsum("ni,ni->n", n1, n2))
    assert d.max() > 1e-8  # Q2 visibly perturbs the normals ...
    _, on, _ = O.covariances(far, None, 20, 1e-3)
    assert_normals_match(n1, on, far, nbr)  # ... and the default matches the reference


# ------------------------------------------------------------------------------------------------
# BASELINE metric size (100K x 100K): size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def lidar100k():
    return synth.lidar_pair(seed=2, n_points=100_000)


def test_metric_size_properties(lidar100k, nn, lm):
    src, sl, tgt, tl, T_gt, cm = lidar100k
    e, p = make_engine(sicp.MODE_EM, 11, cm, nn_method=nn, lm_on_device=lm)
    e.set_source(src, sl)
    e.set_target(tgt, tl)
    # kNN: sorted, gated, self-consistent distances; spot-check rows against the oracle kd-tree
    qt0 = mat_to_qt(synth.pose_matrix(0.5, (0, 0, 1), (0.2, 0.1, 0.0)))
    idx, d2, w = e.correspondences(qt0)
    assert idx.shape == (100_000, 4) and (np.diff(d2, axis=1) >= 0).all()
    assert ((idx >= 0) == (d2 < np.float32(250))).all()
    q = O.transform_points(O.se3_matrix(qt0), src)
    rows = np.random.default_rng(0).choice(100_000, 2000, replace=False)
    oi, od = O.knn(q[rows], tgt, 4, kdtree=True)
    oi[~(od < np.float32(250))] = -1
    assert np.array_equal(idx[rows], oi) and np.array_equal(d2[rows], od)
    dd = q[:, None, :] - tgt[np.maximum(idx, 0)]
    rec = (dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]) + dd[..., 2] * dd[..., 2]
    assert np.array_equal(rec[idx >= 0], d2[idx >= 0])
    assert (w >= 0).all() and (w[idx < 0] == 0).all() and w.max() <= 1.0 + 1e-12
    # H symmetric positive definite, deterministic
    a1, a2 = e.accumulate(qt0), e.accumulate(qt0)
    assert np.array_equal(a1, a2)
    H = np.zeros((6, 6)); H[np.triu_indices(6)] = a1[:21]; H = H + H.T - np.diag(np.diag(H))
    assert np.linalg.eigvalsh(H).min() > 0
    # planted transform recovery, idempotence, determinism
    qt, st = e.align()
    rot, tr = pose_err_to_matrix(qt, T_gt)
    assert rot < 2e-3 and tr < 2e-2, (rot, tr)
    qt_b, st_b = e.align()
    assert np.array_equal(qt, qt_b) and st["total_evals"] == st_b["total_evals"]
    qt2, st2 = e.align(qt)
    rot, tr = pose_delta(qt, qt2)
    assert st2["outer_iters"] == 1 and rot < 3.2e-3 and tr < 3.2e-3  # inside the outer stop radius sqrt(1e-5)


# ------------------------------------------------------------------------------------------------
# lock-step batch (sicp_align_batch): per pair bit-identical to sicp_align, and equal to the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["em", "gicp", "semantic"])
def test_align_batch_equals_single_align(mode):
    m = {"em": sicp.MODE_EM, "gicp": sicp.MODE_GICP, "semantic": sicp.MODE_SEMANTIC}[mode]
    pairs = []
    # pairs of different size and difficulty: they converge after different numbers of outer iterations
    for seed, n in ((2, 6000), (3, 2500), (5, 9000)):
        ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=seed, n_points=n)
        pairs.append((ps, ls, pt, lt, cm))
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    engines, singles = [], []
    try:
        for ps, ls, pt, lt, cm in pairs:
            e, p = make_engine(m, 11 if mode == "em" else 0, cm if mode == "em" else None)
            e.set_source(ps, ls if mode != "gicp" else None)
            e.set_target(pt, lt if mode != "gicp" else None)
            engines.append(e)
            singles.append(e.align(ident))
        res = sicp.align_batch(engines, np.tile(ident, (len(engines), 1)))
        for (qb, sb), (q1, s1) in zip(res, singles):
            assert np.array_equal(qb, q1)  # same kernels' arithmetic, same block decomposition: same bits
            for key in ("outer_iters", "total_lm_iters", "total_evals", "total_corr", "total_active"):
                assert sb[key] == s1[key], key
        # and a second call on the same handles (graph and buffers reused) gives the same again
        res2 = sicp.align_batch(engines, np.tile(ident, (len(engines), 1)))
        for (qb, _), (q1, _) in zip(res2, singles):
            assert np.array_equal(qb, q1)
        # batch of one, and a sub-batch with another leader
        (q0, _), = sicp.align_batch(engines[1:2])
        assert np.array_equal(q0, singles[1][0])
        r3 = sicp.align_batch(engines[1:])
        assert np.array_equal(r3[0][0], singles[1][0]) and np.array_equal(r3[1][0], singles[2][0])
    finally:
        for e in engines:
            e.close()


def test_accumulate_batch_equals_accumulate():
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    engines, ref, qts = [], [], []
    try:
        for seed, n in ((2, 5000), (4, 12000)):
            ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=seed, n_points=n)
            e, p = make_engine(sicp.MODE_EM, 11, cm)
            e.set_source(ps, ls); e.set_target(pt, lt)
            qt = mat_to_qt(T)
            e.correspondences(ident)
            engines.append(e); qts.append(qt); ref.append(e.accumulate(qt))
        out, ms = sicp.accumulate_batch(engines, np.array(qts))
        assert ms > 0
        for p in range(len(engines)):
            assert np.array_equal(out[p], ref[p])
    finally:
        for e in engines:
            e.close()


def test_align_batch_rejects_mismatched_handles():
    ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=2000)
    e1, _ = make_engine(sicp.MODE_EM, 11, cm)
    e2, _ = make_engine(sicp.MODE_GICP)
    try:
        for e in (e1, e2):
            e.set_source(ps, ls); e.set_target(pt, lt)
        with pytest.raises(RuntimeError):
            sicp.align_batch([e1, e2])
        with pytest.raises(RuntimeError):
            sicp.align_batch([e1, e1])
    finally:
        e1.close(); e2.close()


def test_align_batch_more_pairs_than_one_job_launch():
    """10 pairs: 20 covariance jobs and 10 search jobs per phase exceed one job-array launch (16 / 8)."""
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    engines, singles = [], []
    try:
        for k in range(10):
            ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=10 + k, n_points=1500 + 100 * k)
            e, p = make_engine(sicp.MODE_EM, 11, cm)
            e.set_source(ps, ls); e.set_target(pt, lt)
            engines.append(e)
            singles.append(e.align(ident))
        res = sicp.align_batch(engines)
        for (qb, sb), (q1, s1) in zip(res, singles):
            assert np.array_equal(qb, q1) and sb["outer_iters"] == s1["outer_iters"] and sb["total_active"] == s1["total_active"]
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("knob", [dict(nn_method=2), dict(nn_method=0), dict(profile=1)])
def test_align_batch_per_pair_launch_path(knob):
    """With another search engine or with profiling on, a batch keeps the pairs' own searches and
    feature kernels on their own streams (only the inner solves are batched): same results."""
    engines, singles = [], []
    try:
        for seed, n in ((2, 3000), (7, 4500)):
            ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=seed, n_points=n)
            e, p = make_engine(sicp.MODE_EM, 11, cm, **knob)
            e.set_source(ps, ls); e.set_target(pt, lt)
            engines.append(e)
            singles.append(e.align())
        for (qb, sb), (q1, s1) in zip(sicp.align_batch(engines), singles):
            assert np.array_equal(qb, q1) and sb["outer_iters"] == s1["outer_iters"] and sb["total_lm_iters"] == s1["total_lm_iters"]
    finally:
        for e in engines:
            e.close()


def test_align_batch_with_tiny_and_ragged_pairs():
    """A batch mixing a one-leaf cloud, a cloud whose size is not a multiple of the leaf / group sizes
    and a normal one: every job kernel sees q_count and tree depths that differ inside one launch."""
    rng = np.random.default_rng(5)
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    engines, singles = [], []
    try:
        for n in (25, 333, 4000, 61):
            tgt = rng.uniform(0, 4, (n, 3)).astype(np.float32)
            tgt[:, 2] *= 0.05  # a slab: well-conditioned normals
            R = Rotation.from_rotvec([0.01, -0.02, 0.015]).as_matrix()
            src = ((tgt.astype(np.float64) - [0.03, 0.01, 0.0]) @ R).astype(np.float32)
            e, p = make_engine(sicp.MODE_GICP)
            e.set_source(src, None); e.set_target(tgt, None)
            engines.append(e)
            singles.append(e.align(ident))
        for (qb, sb), (q1, s1) in zip(sicp.align_batch(engines), singles):
            assert np.array_equal(qb, q1) and sb["outer_iters"] == s1["outer_iters"]
    finally:
        for e in engines:
            e.close()


def test_knn_with_seed_hints_is_still_exact():
    """The second and later searches of the same queries start from the previous result (seed hint,
    trusted only while close).  Whatever the seed, the result must stay bit-identical to the kd-tree."""
    ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=9, n_points=20000)
    e, p = make_engine(sicp.MODE_EM, 11, cm)
    try:
        e.set_source(ps, ls); e.set_target(pt, lt)
        poses = [np.array([0, 0, 0, 1, 0, 0, 0.0]), mat_to_qt(synth.pose_matrix(0.2, (0, 0, 1), (0.02, 0.0, 0.0))),
                 mat_to_qt(T), mat_to_qt(synth.pose_matrix(25.0, (0, 0, 1), (3.0, -2.0, 0.5))),  # far jump: stale hints
                 mat_to_qt(T)]
        for qt in poses:
            idx, d2, _ = e.correspondences(qt)
            q = O.transform_points(O.se3_matrix(qt), ps)
            oi, od = O.knn(q, pt, 4)
            want = np.where(od < np.float32(250), oi, -1)
            assert np.array_equal(idx, want) and np.array_equal(d2, od)
    finally:
        e.close()


def test_align_batch_at_metric_size_equals_single():
    ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=100000)
    engines = []
    try:
        for k in range(4):
            e, p = make_engine(sicp.MODE_EM, 11, cm)
            e.set_source(ps, ls); e.set_target(pt, lt)
            engines.append(e)
        q1, s1 = engines[0].align()
        for qb, sb in sicp.align_batch(engines):
            assert np.array_equal(qb, q1) and sb["outer_iters"] == s1["outer_iters"] and sb["total_lm_iters"] == s1["total_lm_iters"]
    finally:
        for e in engines:
            e.close()


def test_align_batch_of_sixty_pairs_one_tick_group():
    """More than 48 pairs: the batch runs as ONE tick group (below it, two alternate).  Same bits as
    lone aligns either way; 60 small GICP pairs of ragged sizes."""
    rng = np.random.default_rng(11)
    engines, singles = [], []
    try:
        for k in range(60):
            n = int(rng.integers(300, 1500))
            tgt = rng.uniform(0, 6, (n, 3)).astype(np.float32)
            tgt[:, 2] = (0.1 * np.sin(tgt[:, 0]) + 0.05 * tgt[:, 1]).astype(np.float32)  # a gently curved sheet
            R = Rotation.from_rotvec(rng.normal(0, 0.01, 3)).as_matrix()
            src = ((tgt.astype(np.float64) - rng.normal(0, 0.02, 3)) @ R).astype(np.float32)
            e, p = make_engine(sicp.MODE_GICP)
            e.set_source(src, None); e.set_target(tgt, None)
            engines.append(e)
            singles.append(e.align())
        for k, ((qb, sb), (q1, s1)) in enumerate(zip(sicp.align_batch(engines), singles)):
            assert np.array_equal(qb, q1), k
            assert sb["outer_iters"] == s1["outer_iters"] and sb["total_evals"] == s1["total_evals"]
    finally:
        for e in engines:
            e.close()


def test_align_batch_of_more_pairs_than_one_launch_holds():
    """A launch evaluates at most 256 pairs: in a batch of 280 the others wait with their search done
    and join as slots free up.  Same bits and counters as lone aligns."""
    rng = np.random.default_rng(12)
    engines, singles = [], []
    try:
        for k in range(280):
            n = int(rng.integers(300, 1200))
            tgt = rng.uniform(0, 6, (n, 3)).astype(np.float32)
            tgt[:, 2] = (0.1 * np.sin(tgt[:, 0]) + 0.05 * tgt[:, 1]).astype(np.float32)
            R = Rotation.from_rotvec(rng.normal(0, 0.01, 3)).as_matrix()
            src = ((tgt.astype(np.float64) - rng.normal(0, 0.02, 3)) @ R).astype(np.float32)
            e, p = make_engine(sicp.MODE_GICP)
            e.set_source(src, None); e.set_target(tgt, None)
            engines.append(e)
            singles.append(e.align())
        for k, ((qb, sb), (q1, s1)) in enumerate(zip(sicp.align_batch(engines), singles)):
            assert np.array_equal(qb, q1), k
            assert sb["outer_iters"] == s1["outer_iters"] and sb["total_evals"] == s1["total_evals"]
    finally:
        for e in engines:
            e.close()


def test_accumulate_batch_with_long_workgroup_ranges():
    """48 pairs x 60K points x K = 4: 5640 chunks over the 512 persistent workgroups = 11 per workgroup --
    ranges that span pair boundaries (several segments) and exceed the 8 chunks whose wave sums are
    parked in LDS between two combining passes.  Every pair's 28 sums equal its lone evaluation."""
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=6, n_points=60000)
    rng = np.random.default_rng(13)
    engines, ref, qts = [], [], []
    try:
        for k in range(48):
            e, p = make_engine(sicp.MODE_EM, 11, cm)
            if k == 0:
                e.set_source(ps, ls); e.set_target(pt, lt)
            else:  # the same two clouds, shared: 48 handles, 2 clouds
                e.share_cloud(sicp.SOURCE, engines[0], sicp.SOURCE); e.share_cloud(sicp.TARGET, engines[0], sicp.TARGET)
            e.correspondences(ident)
            qt = mat_to_qt(T)
            qt[4:] += rng.normal(0, 0.01, 3)  # every pair is evaluated at its own pose
            engines.append(e); qts.append(qt); ref.append(e.accumulate(qt))
        out, ms = sicp.accumulate_batch(engines, np.array(qts))
        for p in range(len(engines)):
            assert np.array_equal(out[p], ref[p]), p
        assert len({tuple(r) for r in ref}) == 48
    finally:
        for e in engines:
            e.close()

"""GPU tests (-m gpu) of the branches and entry points around the hot path that the parity suite does
not reach: the device build of SE(3), the non-quirk Probability branch, any covariance k, shared
clouds / feature reuse, device-resident uploads, and argument validation.  All through the C ABI.
"""
import importlib
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import oracle_lib as O
import synth
from np_ref import mat_to_qt

pytestmark = pytest.mark.gpu

sicp = importlib.import_module("semantic-icp_amd")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IDENT = np.array([0, 0, 0, 1, 0, 0, 0.0])


def make_engine(mode, C=0, cm=None, **kw):
    p = sicp.default_params(mode)
    p.num_classes = C
    for k, v in kw.items():
        setattr(p, k, v)
    e = sicp.Engine(0, p)
    if cm is not None:
        e.set_confusion(cm)
    return e


def oracle_params(mode, C=0, **kw):
    p = O.default_params(mode)
    p.num_classes = C
    p.use_kdtree = 1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def pose_delta(qa, qb):
    D = np.linalg.inv(O.se3_matrix(qa)) @ O.se3_matrix(qb)
    return np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()), np.linalg.norm(D[:3, 3])


# ------------------------------------------------------------------------------------------------
# a8: SE(3) exp / log / plus / mul / inverse as compiled for the device (csrc/se3.hpp)
# ------------------------------------------------------------------------------------------------
def test_se3_device_against_expm_logm_golden():
    g = np.load(os.path.join(G, "se3.npz"))
    A, B = g["A"], g["B"]
    e = sicp.Engine(0)
    try:
        qa = e.se3_device(sicp.SE3_EXP, A)
        qb = e.se3_device(sicp.SE3_EXP, B)
        for i in range(len(A)):
            assert np.allclose(O.se3_matrix(qa[i]), g["exp_mats"][i], atol=1e-14, rtol=0)
        assert np.allclose(e.se3_device(sicp.SE3_LOG, qa), A, atol=1e-13, rtol=0)
        prod = e.se3_device(sicp.SE3_MUL, np.concatenate([qa, qb], axis=1))
        plus = e.se3_device(sicp.SE3_PLUS, np.concatenate([qa, B], axis=1))
        inv = e.se3_device(sicp.SE3_INV, qa)
        for i in range(len(A)):
            assert np.allclose(O.se3_matrix(prod[i]), g["prod"][i], atol=1e-14, rtol=0)
            assert np.allclose(O.se3_matrix(plus[i]), g["prod"][i], atol=1e-14, rtol=0)
            assert np.allclose(O.se3_matrix(inv[i]), g["inv"][i], atol=1e-14, rtol=0)
            assert abs(np.linalg.norm(prod[i][:4]) - 1) < 1e-15
        # the small-angle branches (theta < 1e-10) and the host build of the same header (the oracle is
        # a separate restatement: agreement to the last few ulps, not bits)
        tiny = np.array([[0.3, -0.2, 0.1, 1e-12, -2e-12, 5e-13], [0, 0, 0, 0, 0, 0], [1, 2, 3, 0, 0, 1e-11]])
        qt = e.se3_device(sicp.SE3_EXP, tiny)
        for i in range(len(tiny)):
            assert np.allclose(qt[i], O.se3_exp(tiny[i]), atol=1e-15, rtol=0)
        lg = e.se3_device(sicp.SE3_LOG, qt)
        for i in range(len(tiny)):
            assert np.allclose(lg[i], O.se3_log(qt[i]), atol=1e-15, rtol=0)
        # (Sophus' small-angle exp uses V = R, so the round trip is only exact to |omega| |upsilon|)
        assert np.allclose(lg, tiny, atol=1e-10, rtol=0)
    finally:
        e.close()



def lm_sequences(rng, n, wild):
    """n x (start pose | 24 evaluations x [H upper 21 | g 6 | cost]) for sicp_se3_device(LM_SEQUENCE...): random SPD H
    over ten decades, sometimes indefinite (the Cholesky fails: invalid steps, halved radius), gradients that make steps
    from 1e-9 to whole turns (`wild`) or only small ones, a cost that wanders so that steps are accepted and rejected,
    sometimes a non-finite sum."""
    E = sicp.LM_SEQUENCE_EVALS
    items = np.empty((n, 7 + 28 * E))
    iu = np.triu_indices(6)
    for i in range(n):
        q = rng.normal(size=4)
        items[i, :4] = q / np.linalg.norm(q)
        items[i, 4:7] = rng.normal(scale=10.0, size=3)
        cost = 10.0 ** rng.uniform(-3, 6)
        for e in range(E):
            A = rng.normal(size=(6, 6))
            H = (A @ A.T + 1e-3 * np.eye(6)) * 10.0 ** rng.uniform(-4, 6)
            if rng.uniform() < 0.12:
                j = rng.integers(6)
                H[j, j] = -abs(H[j, j])                       # not positive definite
            g = rng.normal(size=6) * np.sqrt(np.diag(np.abs(H))) * 10.0 ** (rng.uniform(-9, 1) if wild else rng.uniform(-9, -5))
            cost = cost * (rng.uniform(0.3, 1.0) if rng.uniform() < 0.75 else rng.uniform(1.0, 3.0))
            o = np.concatenate([H[iu], g, [cost]])
            r = rng.uniform()
            if r < 0.02:
                o[rng.integers(28)] = np.nan
            elif r < 0.03:
                o[27] = np.inf
            items[i, 7 + 28 * e: 7 + 28 * (e + 1)] = o
    return items


def test_lm_step_by_a_wave_equals_the_one_lane_machine_on_any_sequence(tmp_path):
    """The kernels step the trust-region machine with a whole wavefront (lm_feed<true>: the six sqrt(diag / radius) and the two
    sincos of a Plus in different lanes, the finite test by ballot); the host loop and the oracle's restatement run it in one
    lane.  The machine is a pure function of (state, evaluation), so the two forms must agree BIT FOR BIT on any sequence of
    evaluations, also on ones no registration produces: indefinite H (failed factorisations, invalid steps), rejected
    steps, non-finite sums, steps of whole turns.  And on sequences with small rotations -- where the host's libm and the
    device's agree on sin and cos -- both must equal csrc/lm.hpp compiled for the host (what lm_on_device = 0 runs)."""
    import subprocess
    rng = np.random.default_rng(20261003)
    wild = lm_sequences(rng, 3000, True)
    small = lm_sequences(rng, 1000, False)
    with make_engine(sicp.MODE_GICP) as e:
        w_wave, w_lane = e.se3_device(sicp.LM_SEQUENCE, wild), e.se3_device(sicp.LM_SEQUENCE_ONE_LANE, wild)
        s_wave, s_lane = e.se3_device(sicp.LM_SEQUENCE, small), e.se3_device(sicp.LM_SEQUENCE_ONE_LANE, small)
    assert w_wave.tobytes() == w_lane.tobytes()
    assert s_wave.tobytes() == s_lane.tobytes()
    # the sequences reach the branches they are meant to reach
    status, iters, evals, invalid = w_wave[:, 31], w_wave[:, 32], w_wave[:, 33], w_wave[:, 34]
    assert set(np.unique(status).astype(int)) >= {-1, 0, 3}             # still running / converged / failed first evaluation
    assert (iters > evals).any() and (evals > 3).any()                   # retries inside one feed (invalid steps); long runs
    assert 2 in set(np.unique(status).astype(int)) or (invalid > 0).any()
    # the host build of the same header on the small-rotation sequences
    src = tmp_path / "lm_seq.cc"
    src.write_text(r"""
#include <cstdio>
#include <vector>
#define SICP_HD
#include "lm.hpp"
using namespace sicp;
int main(int argc, char** argv) {
  const int E = 24, IN = 7 + 28 * E, OUT = 37;
  FILE* f = std::fopen(argv[1], "rb"); int n = 0; if (!f || std::fread(&n, 4, 1, f) != 1) return 2;
  std::vector<double> in((size_t)n * IN), out((size_t)n * OUT);
  if (std::fread(in.data(), 8, in.size(), f) != in.size()) return 2; std::fclose(f);
  for (int i = 0; i < n; ++i) {
    const double* item = &in[(size_t)i * IN];
    LmState s; LmOptions opt; lm_init(s, opt, item);
    for (int e = 0; e < E && s.status == LM_RUNNING; ++e) lm_feed(s, item + 7 + 28 * e);
    double* r = &out[(size_t)i * OUT]; int k = 0;
    for (int j = 0; j < 7; ++j) r[k++] = s.pose[j];
    for (int j = 0; j < 7; ++j) r[k++] = s.x[j];
    for (int j = 0; j < 6; ++j) r[k++] = s.diag[j];
    for (int j = 0; j < 6; ++j) r[k++] = s.scale[j];
    r[k++] = s.radius; r[k++] = s.cost; r[k++] = s.model_change; r[k++] = s.decrease_factor; r[k++] = s.x_norm;
    r[k++] = s.status; r[k++] = s.iterations; r[k++] = s.evaluations; r[k++] = s.invalid; r[k++] = s.reuse_diagonal; r[k++] = s.phase;
  }
  f = std::fopen(argv[2], "wb"); std::fwrite(out.data(), 8, out.size(), f); std::fclose(f);
  return 0;
}
""")
    exe = tmp_path / "lm_seq"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(root, "semantic-icp_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.int32(small.shape[0]).tobytes())
        f.write(small.tobytes())
    subprocess.run([str(exe), str(fin), str(fout)], check=True)
    host = np.fromfile(fout).reshape(small.shape[0], sicp.LM_SEQUENCE_OUT)
    same = np.all((host == s_wave) | (np.isnan(host) & np.isnan(s_wave)), axis=1)
    assert same.all(), f"{(~same).sum()} of {same.size} sequences differ between the host and the device build; first: {host[~same][0]} / {s_wave[~same][0]}"

# ------------------------------------------------------------------------------------------------
# a6: GICPCostFunction::Probability as a double (quirk Q1 switched off)
# ------------------------------------------------------------------------------------------------
def test_probability_as_double_branch():
    src, sl, tgt, tl, T_gt = synth.config1_pair(seed=1, n_per_label=400)
    C = 4
    cm = synth.confusion_matrix(C)
    qt = mat_to_qt(synth.pose_matrix(1.0, (0, 1, 0), (0.05, 0.0, -0.02)))
    e1 = make_engine(sicp.MODE_EM, C, cm)                             # reference behaviour: bool
    e0 = make_engine(sicp.MODE_EM, C, cm, quirk_bool_probability=0)   # the double it was meant to be
    try:
        for e in (e0, e1):
            e.set_source(src, sl); e.set_target(tgt, tl)
        idx, d2, w0 = e0.correspondences(qt)
        idx1, _, w1 = e1.correspondences(qt)
        assert np.array_equal(idx, idx1)
        scov, _, sh = O.covariances(src, sl, 20, 1e-3, C)
        tcov, _, th = O.covariances(tgt, tl, 20, 1e-3, C)
        want0 = np.zeros(idx.shape); want1 = np.zeros(idx.shape)
        for i in range(len(src)):
            for c in range(idx.shape[1]):
                j = idx[i, c]
                if j < 0:
                    continue
                b, v = O.gicp_probability(qt, src[i].astype(np.float64), tgt[j].astype(np.float64), scov[i], tcov[j])
                prob = O.em_prob(cm, th[j], sh[i])
                want0[i, c] = prob * v
                want1[i, c] = prob * float(b)
        assert np.allclose(w1, want1, rtol=1e-12, atol=0)
        # pow() / exp() of the device library vs libm: a few ulps
        assert np.allclose(w0, want0, rtol=1e-11, atol=1e-300)
        assert (w0 <= w1 + 1e-15).all() and np.abs(w0 - w1).max() > 1e-3  # the branch really changes the weights
        # fused labels with the double: against the oracle?  The oracle implements the reference (bool) only;
        # check the documented property instead: the arg-max uses the same weights the kernel above produced
        lab0, lab1 = e0.fused_labels(qt), e1.fused_labels(qt)
        assert lab0.shape == lab1.shape and lab0.min() >= 1 and lab0.max() <= C
        # and the solve still converges to the planted pose
        q0, st0 = e0.align()
        D = np.linalg.inv(T_gt) @ O.se3_matrix(q0)
        assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < 5e-3 and np.linalg.norm(D[:3, 3]) < 3e-2
    finally:
        e0.close(); e1.close()


# ------------------------------------------------------------------------------------------------
# a13: any constructor k (em_icp.h:42, gicp.h:34, semantic_point_cloud.h:31)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [2, 5, 12, 19, 21, 27, 32])
@pytest.mark.parametrize("nn_method", [0, 1, 2], ids=["bruteforce", "boxtree", "boxtree_per_query"])
def test_any_covariance_k(k, nn_method):
    src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=4, n_points=6000)
    e = make_engine(sicp.MODE_EM, 11, cm, k_cov=k, nn_method=nn_method)
    try:
        e.set_source(src, sl)
        cov, nrm, hist, nbr = e.covariances(sicp.SOURCE, want_hist=True, want_nn=True)
        onn, _ = O.knn(src, src, k, kdtree=True)
        assert nbr.shape == (len(src), k) and np.array_equal(nbr, onn)        # bit-exact lists
        want_hist = np.stack([np.bincount(sl[row] - 1, minlength=11) for row in onn]).astype(np.uint8)
        assert np.array_equal(hist, want_hist)
        ocov, onrm, ohist = O.covariances(src, sl, k, 1e-3, 11, kdtree=True)
        assert np.allclose(hist / float(k), ohist, atol=1e-15)
        if k >= 5:
            dots = np.abs(np.einsum("ni,ni->n", nrm, onrm))
            assert np.median(1 - dots) < 1e-12
    finally:
        e.close()


@pytest.mark.parametrize("mode", [sicp.MODE_GICP, sicp.MODE_EM])
def test_align_with_k10_vs_oracle(mode):
    src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=6, n_points=8000)
    lab = mode == sicp.MODE_EM
    e = make_engine(mode, 11 if lab else 0, cm if lab else None, k_cov=10)
    try:
        e.set_source(src, sl if lab else None); e.set_target(tgt, tl if lab else None)
        qt, st = e.align()
        oq, ost = O.align(oracle_params(mode, 11 if lab else 0, k_cov=10), src, sl if lab else None, tgt, tl if lab else None,
                          cm if lab else None, IDENT)
        rot, tr = pose_delta(qt, oq)
        assert rot < 1e-7 and tr < 1e-7 and st["outer_iters"] == ost["outer_iters"] and st["total_active"] == ost["total_active"]
    finally:
        e.close()


def test_rejected_k_names_the_supported_range():
    e = sicp.Engine(0)
    try:
        for field, bad in (("k_cov", 0), ("k_cov", 33), ("k_cov", -4), ("knn", 3), ("knn", 0)):
            p = sicp.default_params(sicp.MODE_EM)
            p.num_classes = 3
            setattr(p, field, bad)
            with pytest.raises(sicp.SicpError) as err:
                e.set_params(p)
            assert err.value.status == sicp.ERR_INVALID_ARGUMENT
            assert ("1..32" in str(err.value)) if field == "k_cov" else ("1, 4 or 20" in str(err.value))
        p = sicp.default_params(sicp.MODE_EM)
        p.num_classes = 3
        p.k_cov = 32
        e.set_params(p)  # the upper end is accepted
    finally:
        e.close()


# ------------------------------------------------------------------------------------------------
# b: sicp_set_cloud_device
# ------------------------------------------------------------------------------------------------
class HipBuffers:
    """Device buffers through the HIP runtime libsicp.so itself links (torch would bring a second copy)."""

    def __init__(self):
        import ctypes as C

        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.ptrs = []

    def upload(self, arr):
        C = self.C
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), C.c_size_t(arr.nbytes)) == 0
        assert self.hip.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0  # hipMemcpyHostToDevice
        self.ptrs.append(p)
        return p.value

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []


def test_set_cloud_device_equals_host_upload():
    src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=3, n_points=5000)
    e_host = make_engine(sicp.MODE_EM, 11, cm)
    e_dev = make_engine(sicp.MODE_EM, 11, cm)
    g = make_engine(sicp.MODE_GICP)
    g2 = make_engine(sicp.MODE_GICP)
    dev = HipBuffers()
    try:
        e_host.set_source(src, sl); e_host.set_target(tgt, tl)
        addr = {}
        for which, xyz, lab in ((sicp.SOURCE, src, sl), (sicp.TARGET, tgt, tl)):
            addr[which] = [dev.upload(xyz[:, i].astype(np.float32)) for i in range(3)] + [dev.upload(lab.astype(np.uint32))]
            e_dev.set_cloud_device(which, len(xyz), *addr[which])
        qh, sh = e_host.align()
        qd, sd = e_dev.align()
        assert np.array_equal(qh, qd) and sh["outer_iters"] == sd["outer_iters"] and sh["total_active"] == sd["total_active"]
        # no labels: GICP
        g.set_cloud_device(sicp.SOURCE, len(src), *addr[sicp.SOURCE][:3], None)
        g.set_cloud_device(sicp.TARGET, len(tgt), *addr[sicp.TARGET][:3], None)
        g2.set_source(src); g2.set_target(tgt)
        assert np.array_equal(g.align()[0], g2.align()[0])
    finally:
        for x in (e_host, e_dev, g, g2):
            x.close()
        dev.free()


# ------------------------------------------------------------------------------------------------
# setSourceCloud(cloud, kdtree, covs): shared clouds and feature reuse (gicp.h:48-56, kitti_eval.cc:207-226)
# ------------------------------------------------------------------------------------------------
def sequence(n_scans=5, n_points=6000):
    scans, poses, cm = synth.lidar_sequence(seed=5, n_scans=n_scans, n_points=n_points, step=(1.0, 2.0))
    return scans, poses, cm


@pytest.mark.parametrize("mode", ["em", "gicp"])
@pytest.mark.parametrize("reuse", [0, 1])
def test_shared_clouds_in_a_sequence_batch(mode, reuse):
    """Pair p registers scan p+1 onto scan p; scan p+1 is uploaded once and shared as the target of
    pair p+1.  Results must equal separately uploaded, lone aligns bit for bit."""
    m = sicp.MODE_EM if mode == "em" else sicp.MODE_GICP
    lab = mode == "em"
    scans, poses, cm = sequence()
    n_pairs = len(scans) - 1
    lone, shared = [], []
    try:
        want = []
        for p in range(n_pairs):
            e = make_engine(m, 11 if lab else 0, cm if lab else None)
            e.set_source(scans[p + 1][0], scans[p + 1][1] if lab else None)
            e.set_target(scans[p][0], scans[p][1] if lab else None)
            lone.append(e)
            want.append(e.align())
        for p in range(n_pairs):
            e = make_engine(m, 11 if lab else 0, cm if lab else None, reuse_features=reuse)
            e.set_source(scans[p + 1][0], scans[p + 1][1] if lab else None)
            if p == 0:
                e.set_target(scans[0][0], scans[0][1] if lab else None)
            else:
                e.share_cloud(sicp.TARGET, shared[p - 1], sicp.SOURCE)
            shared.append(e)
        # lock step
        res = sicp.align_batch(shared)
        for (qb, sb), (q1, s1) in zip(res, want):
            assert np.array_equal(qb, q1)
            for key in ("outer_iters", "total_lm_iters", "total_evals", "total_corr", "total_active"):
                assert sb[key] == s1[key], key
        # again (with reuse_features the features are not recomputed at all), and one after the other
        res = sicp.align_batch(shared)
        for (qb, _), (q1, _) in zip(res, want):
            assert np.array_equal(qb, q1)
        for e, (q1, s1) in zip(shared, want):
            qb, sb = e.align()
            assert np.array_equal(qb, q1) and sb["outer_iters"] == s1["outer_iters"]
        # a new upload into a shared slot leaves the other handle's cloud alone
        shared[1].set_target(scans[0][0], scans[0][1] if lab else None)
        qb, _ = shared[0].align()
        assert np.array_equal(qb, want[0][0])
        e2 = make_engine(m, 11 if lab else 0, cm if lab else None)
        e2.set_source(scans[2][0], scans[2][1] if lab else None)
        e2.set_target(scans[0][0], scans[0][1] if lab else None)
        assert np.array_equal(shared[1].align()[0], e2.align()[0])
        e2.close()
    finally:
        for e in lone + shared:
            e.close()


def test_reuse_features_skips_the_covariance_search():
    src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=8, n_points=5000)
    e = make_engine(sicp.MODE_EM, 11, cm, reuse_features=1, profile=2)  # SICP_PROFILE_COV: count the self-searches
    r = make_engine(sicp.MODE_EM, 11, cm, profile=2)
    try:
        for x in (e, r):
            x.set_source(src, sl); x.set_target(tgt, tl)
        q1, s1 = e.align()
        q2, s2 = e.align()
        q3, s3 = r.align()
        q4, s4 = r.align()
        assert np.array_equal(q1, q2) and np.array_equal(q1, q3) and np.array_equal(q3, q4)
        assert s1["cov_launches"] == 2 and s2["cov_launches"] == 0       # kept
        assert s3["cov_launches"] == 2 and s4["cov_launches"] == 2       # the reference recomputes (em_icp.hpp:28-29)
        e.set_source(src, sl)                                            # a new upload invalidates
        _, s5 = e.align()
        assert s5["cov_launches"] == 1
        p = e.get_params(); p.k_cov = 12; e.set_params(p)                # so does another k
        _, s6 = e.align()
        assert s6["cov_launches"] == 2
    finally:
        e.close(); r.close()


def test_share_cloud_argument_checks():
    a, b = sicp.Engine(0), sicp.Engine(0)
    try:
        with pytest.raises(sicp.SicpError) as err:
            b.share_cloud(sicp.TARGET, a, sicp.SOURCE)  # nothing uploaded yet
        assert err.value.status == sicp.ERR_NOT_READY
        with pytest.raises(sicp.SicpError):
            b.share_cloud(5, a, sicp.SOURCE)
    finally:
        a.close(); b.close()

"""The EM weight kernel alone (HIP events, SICP_PROFILE_WEIGHT) on one 100K x 100K pair at the identity and at the planted pose.
usage (GPU box): [SICP_WEIGHTS_FROM_HIST=1] bench_weights.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from np_ref import mat_to_qt
sicp = importlib.import_module("semantic-icp_amd")
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=100000)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11; p.profile = 4
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
with sicp.Engine(0, p) as e:
    e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt)
    for q in (ident, mat_to_qt(T)):
        e.correspondences(q)
        s0 = e.stats()
        for _ in range(30):
            idx, d2, w = e.correspondences(q)
        s1 = e.stats()
        print(f"{'hist' if os.environ.get('SICP_WEIGHTS_FROM_HIST') else 'proj'}: weight kernel {1e3 * (s1['weight_kernel_ms'] - s0['weight_kernel_ms']) / (s1['weight_launches'] - s0['weight_launches']):.2f} us "
              f"(live slots {(idx >= 0).mean():.3f}, checksum {w.sum():.12e})")
    cov, nrm, hist, nn = e.covariances(sicp.TARGET, want_hist=True)
    nz = (hist > 0).sum(axis=1)
    print("labels present per point: mean %.2f, pure %.3f, <=2 %.3f, <=3 %.3f" % (nz.mean(), (nz == 1).mean(), (nz <= 2).mean(), (nz <= 3).mean()))
    dev = hist.reshape(-1, 64, hist.shape[1]) if len(hist) % 64 == 0 else hist[: len(hist) // 64 * 64].reshape(-1, 64, hist.shape[1])
    print("labels present per 64 consecutive caller-order points (not the device order): mean %.2f" % ((dev > 0).any(axis=1).sum(axis=1).mean()))

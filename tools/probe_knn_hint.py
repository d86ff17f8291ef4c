"""Time the K=4 search alone: first call (curve-position seed) and repeated calls (hint seed), at the
identity (far from aligned) and at the converged pose."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=100000)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11; p.profile = 1
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
with sicp.Engine(0, p) as e:
    e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt)
    q, st = e.align(ident)
    for name, pose in (("identity", ident), ("converged", q)):
        for rep in range(4):
            e.set_source(ps, ls)  # drops the hint
            b = e.stats(); e.correspondences(pose); a = e.stats()
            t_first = a["nn_kernel_ms"] - b["nn_kernel_ms"]
            b = e.stats(); e.correspondences(pose); a = e.stats()
            t_hint = a["nn_kernel_ms"] - b["nn_kernel_ms"]
        print(f"{name}: first (no hint) {1e3 * t_first:.1f} us, with hint {1e3 * t_hint:.1f} us")

"""Kernel-time probe on a small pair (few accumulate blocks): isolates the fixed cost of
lm_step_kernel (state load/store + lm_feed) from its partial-sum reduction."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=n)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = cm.shape[0]
with sicp.Engine(0, p) as e:
    e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt)
    for _ in range(5):
        q, st = e.align()
    print(q, st.outer_iters, st.lm_iters)

import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=n)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
es = []
for k in range(2):
    e = sicp.Engine(0, p); e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt); es.append(e)
q1, s1 = es[0].align()
print("single", q1, {k: s1[k] for k in ("outer_iters", "total_lm_iters", "total_evals", "total_corr", "t_total_ms", "t_solve_ms", "t_cov_ms")})
for rep in range(2):
    t0 = time.perf_counter()
    res = sicp.align_batch(es)
    dt = time.perf_counter() - t0
    for q, s in res:
        print("batch", dt * 1e3, q, {k: s[k] for k in ("outer_iters", "total_lm_iters", "total_evals", "total_corr", "t_total_ms", "t_solve_ms", "t_cov_ms")})

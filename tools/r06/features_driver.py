"""Driver for the PMC passes over the per-align kernels that are not the accumulate / search kernels: 16 different 100K x 100K
EM-ICP pairs through sicp_align_batch (a batch of more than 4 pairs: the EM weights run as em_weight_rows4_jobs_kernel, the
covariances as cov_jobs_kernel, the projections as proj_rows_jobs_kernel -- the kernels of bench.py's timed region).
usage (GPU box): features_driver.py [pairs] [points] [batches]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sicp = importlib.import_module("semantic-icp_amd")
cm = synth.confusion_matrix(11)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
es = []
for k in range(S):
    seed = 2 + k
    motion = (1.0, 2.0) if seed == 2 else (0.5 + 0.11 * (seed % 11), -2.6 + 0.65 * (seed % 9))
    ps, ls, pt, lt, T, _ = synth.lidar_pair(seed=seed, n_points=n, motion=motion)
    e = sicp.Engine(0, p); e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt); es.append(e)
for _ in range(reps):
    res = sicp.align_batch(es)
print("outer iterations:", [st["outer_iters"] for _, st in res])
for e in es:
    e.close()

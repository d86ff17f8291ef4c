"""One pair alone (sicp_align, the reference's call pattern): ms per align() for three 100K x 100K pairs.
usage (GPU box): [SICP_KNN_SPLIT_RATIO=r] one_pair_latency.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
out = []
for seed in (2, 5, 9):
    src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=seed, n_points=100000)
    p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
    with sicp.Engine(0, p) as e:
        e.set_confusion(cm); e.set_source(src, sl); e.set_target(tgt, tl)
        q0, st0 = e.align(ident)
        e.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            q, st = e.align(ident)
        dt = (time.perf_counter() - t) / 10
        out.append((seed, round(1e3 * dt, 3), st["outer_iters"], st["total_evals"], q.tobytes().hex()[:24]))
print("split ratio", os.environ.get("SICP_KNN_SPLIT_RATIO", "default"), out, flush=True)

"""K=4 search time against the number of queries (same 100K-point target): is the launch bound by
throughput (time ~ queries) or by its longest packet (time flat)?  Development aid."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=100000)
big_s, big_l = np.concatenate([ps, ps + 0.01, ps - 0.01, ps + 0.02]), np.concatenate([ls] * 4)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11; p.profile = 1
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
rng = np.random.default_rng(1)
for n in (6250, 12500, 25000, 50000, 100000, 200000, 400000):
    sel = np.sort(rng.choice(len(big_s), n, replace=False)) if n < len(big_s) else np.arange(len(big_s))
    with sicp.Engine(0, p) as e:
        e.set_confusion(cm); e.set_source(big_s[sel], big_l[sel]); e.set_target(pt, lt)
        e.correspondences(ident)
        ts = []
        for rep in range(5):
            b = e.stats(); e.correspondences(ident); a = e.stats()
            ts.append(a["nn_kernel_ms"] - b["nn_kernel_ms"])
    print(f"queries {n:7d}: {1e3 * min(ts):8.1f} us (hint seed)   {1e3 * min(ts) / n * 1e3:7.2f} ns/query", flush=True)

"""Where the 11 ms of a drop-in SemanticICP align() at 307 200 points go below the class shim: upload + per-label search trees
(sicp_set_cloud, SICP_MODE_SEMANTIC, 13 label segments), features, the registration itself.  usage (GPU box): semantic_upload_probe.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
src, sl, tgt, tl = synth.rgbd_pair(seed=3)[:4]
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
for mode, name in ((sicp.MODE_SEMANTIC, "semantic"), (sicp.MODE_GICP, "gicp (one segment)")):
    p = sicp.default_params(mode); p.reuse_features = 1
    with sicp.Engine(0, p) as e:
        lab = (sl, tl) if mode == sicp.MODE_SEMANTIC else (None, None)
        rows = []
        for rep in range(6):
            t0 = time.perf_counter(); e.set_source(src, lab[0]); t1 = time.perf_counter(); e.synchronize(); t2 = time.perf_counter()
            e.set_target(tgt, lab[1]); e.synchronize(); t3 = time.perf_counter()
            e.covariances(sicp.SOURCE) if False else None
            q, st = e.align(ident); t4 = time.perf_counter()
            q, st = e.align(ident); t5 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
        r = np.median(np.array(rows[1:]), axis=0) * 1e3
        print(f"{name}: set_source call {r[0]:.2f} ms + wait for upload/tree {r[1]:.2f} ms | set_target incl. wait {r[2]:.2f} ms | first align (features + solve) {r[3]:.2f} ms | "
              f"second align (features kept) {r[4]:.2f} ms | segments {len(np.unique(sl)) if mode == sicp.MODE_SEMANTIC else 1}, outer {st['outer_iters']}")

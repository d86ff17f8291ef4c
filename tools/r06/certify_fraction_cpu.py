"""CPU estimate (scipy cKDTree, float64) of how many K = 4 neighbour sets of outer iteration n+1 are certified unchanged by
the (K+1)-th distance of iteration n and the query's displacement.  Poses per outer iteration come from the oracle
(max_outer cut-offs).  Not a product path: a design probe for DESIGN.md section 3.2 (round 6)."""
import json, sys, os
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O, synth
sys.path.insert(0, ROOT)
import bench

def poses(seed, n):
    src, sl, tgt, tl, T_gt, cm = synth.lidar_pair(seed=seed, n_points=n, motion=bench.pair_motion(seed))
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    out = [ident]
    p = O.default_params(O.MODE_EM); p.num_classes = 11; p.use_kdtree = 1; p.num_threads = 8
    full, st = O.align(p, src, sl, tgt, tl, cm, ident)
    for m in range(st["outer_iters"] - 1):
        p.max_outer = m - 1   # index threshold: stop after m+1 passes
        q, s2 = O.align(p, src, sl, tgt, tl, cm, ident)
        out.append(q)
    return src, tgt, out, st["outer_iters"]

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    res = {}
    for seed in (2, 3, 7, 12):
        src, tgt, qs, outer = poses(seed, n)
        tree = cKDTree(tgt.astype(np.float64))
        rows = []
        prev = None
        for it, q in enumerate(qs):
            M = O.se3_matrix(q)
            qn = (src.astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32).astype(np.float64)
            d, i = tree.query(qn, k=5)
            if prev is not None:
                qo, do, io = prev
                delta = np.linalg.norm(qn - qo, axis=1)
                dn = np.linalg.norm(tgt[io[:, :4]].astype(np.float64) - qn[:, None, :], axis=2).max(axis=1)
                cert = dn < (do[:, 4] - delta) * (1 - 1e-5)
                same = (np.sort(io[:, :4], 1) == np.sort(i[:, :4], 1)).all(1)
                rows.append({"outer": it + 1, "median_delta_m": float(np.median(delta)), "certified": float(cert.mean()),
                             "sets_really_unchanged": float(same.mean()), "order_unchanged": float((io[:, :4] == i[:, :4]).all(1).mean())})
            prev = (qn, d, i)
        res[seed] = {"outer_iters": outer, "per_iteration": rows}
        print(seed, json.dumps(res[seed]), flush=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "r06", "certify_fraction_cpu.json"), "w"), indent=1)
main()

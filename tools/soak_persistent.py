"""Developer aid: the persistent solve against the tick graph, 150 aligns per case (bit equality every time).
usage (GPU box): tools/soak_persistent.py"""
import importlib, sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
cm = synth.confusion_matrix(11)
IDENT = np.array([0, 0, 0, 1, 0, 0, 0.0])
bad = 0
t0 = time.time()
for seed, n in ((51, 100000), (52, 60000), (53, 130000), (54, 3000)):
    src, sl, tgt, tl, T, _ = synth.lidar_pair(seed=seed, n_points=n if n <= 141000 else None)
    src, sl, tgt, tl = src[:n], sl[:n], tgt[:n], tl[:n]
    ref = None
    for mode, C, conf in ((sicp.MODE_EM, 11, cm), (sicp.MODE_GICP, 0, None)):
        p = sicp.default_params(mode); p.num_classes = C; p.lm_on_device = 2
        with sicp.Engine(0, p) as e:
            if conf is not None: e.set_confusion(conf)
            e.set_source(src, sl if C else None); e.set_target(tgt, tl if C else None)
            ref = e.align(IDENT)
        p.lm_on_device = 1
        with sicp.Engine(0, p) as e:
            if conf is not None: e.set_confusion(conf)
            e.set_source(src, sl if C else None); e.set_target(tgt, tl if C else None)
            for r in range(150):
                q, st = e.align(IDENT)
                if not np.array_equal(q, ref[0]) or st["total_evals"] != ref[1]["total_evals"]:
                    bad += 1
                    print("MISMATCH", seed, n, mode, r, flush=True)
        print("ok", seed, n, mode, round(time.time() - t0, 1), flush=True)
print("mismatches", bad)

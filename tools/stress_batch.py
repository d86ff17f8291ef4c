"""Stress: repeated lock-step batches with clouds whose sizes change every round (buffer growth, graph
re-capture, hint invalidation); every result is checked against a lone align() of a fresh engine."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
rng = np.random.default_rng(0)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
cm = synth.confusion_matrix(11)
engines = [sicp.Engine(0, p) for _ in range(6)]
for e in engines:
    e.set_confusion(cm)
bad = 0
for rnd in range(12):
    pairs = []
    for e in engines:
        n = int(rng.integers(800, 30000))
        ps, ls, pt, lt, T, _ = synth.lidar_pair(seed=int(rng.integers(1, 1000)), n_points=n)
        e.set_source(ps, ls); e.set_target(pt, lt)
        pairs.append((ps, ls, pt, lt))
    res = sicp.align_batch(engines)
    k = int(rng.integers(0, len(engines)))
    with sicp.Engine(0, p) as f:
        f.set_confusion(cm); f.set_source(pairs[k][0], pairs[k][1]); f.set_target(pairs[k][2], pairs[k][3])
        q1, s1 = f.align()
    ok = np.array_equal(res[k][0], q1) and res[k][1]["outer_iters"] == s1["outer_iters"]
    bad += not ok
    print(f"round {rnd}: sizes {[len(x[0]) for x in pairs]} check pair {k}: {'ok' if ok else 'MISMATCH'}", flush=True)
for e in engines:
    e.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)

"""Debug aid: lone align() vs sicp_align_batch on the same handles -- counters and pose deltas."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
sicp = importlib.import_module("semantic-icp_amd")
mode = sicp.MODE_GICP
pairs = [synth.lidar_pair(seed=s, n_points=n) for s, n in ((2, 6000), (3, 2500), (5, 9000))]
engines = []
for ps, ls, pt, lt, T, cm in pairs:
    p = sicp.default_params(mode)
    e = sicp.Engine(0, p)
    e.set_source(ps); e.set_target(pt)
    engines.append(e)
singles = [e.align() for e in engines]
singles2 = [e.align() for e in engines]
batch = sicp.align_batch(engines)
batch1 = [sicp.align_batch([e])[0] for e in engines]
keys = ("outer_iters", "total_lm_iters", "total_evals", "total_corr", "total_active", "final_cost")
for k, ((qs, ss), (q2, s2), (qb, sb), (q1, s1)) in enumerate(zip(singles, singles2, batch, batch1)):
    print("pair", k, "single==single2", np.array_equal(qs, q2), "batch==single", np.array_equal(qs, qb), "batch1==single", np.array_equal(qs, q1),
          "max|dq|", np.abs(qs - qb).max(), np.abs(qs - q1).max())
    print("   single", [ss[x] for x in keys])
    print("   batch ", [sb[x] for x in keys], "slots", sb["lockstep_slots"])
    print("   batch1", [s1[x] for x in keys])

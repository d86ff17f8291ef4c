"""The batched accumulate kernel at several launch shapes (pairs per launch), `repeat` launches back to back between
two HIP events (sicp_accumulate_batch): microseconds per pair-evaluation.  4 / 8 / 16 pairs of 100K x 100K points are
52 / 104 / 208 MB of working set: inside the 256 MB Infinity Cache when launched back to back.
usage (GPU box): bench_acc_shapes.py [points] [repeat] > json"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 50
shapes = [1, 2, 4, 8, 16, 32, 64, 128, 256]
pairs = [synth.lidar_pair(seed=2 + k, n_points=n) for k in range(4)]   # four different pairs, cycled (own buffers per engine)
cm = pairs[0][5]
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
es = []
for k in range(max(shapes)):
    ps, ls, pt, lt = pairs[k % 4][:4]
    e = sicp.Engine(0, p); e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt); e.correspondences(ident); es.append(e)
bytes_per_pair = 24 * n + 32 * 4 * n
rows = []
for S in shapes:
    qts = np.tile(ident, (S, 1))
    ms = []
    for rep in range(6):
        out, t = sicp.accumulate_batch(es[:S], qts, repeat=repeat)
        ms.append(t)
    ms = np.array(ms[2:])
    rows.append(dict(pairs_per_launch=S, launch_us=round(1e3 * ms.mean(), 2), launch_us_min=round(1e3 * ms.min(), 2),
                     us_per_pair_evaluation=round(1e3 * ms.mean() / S, 3), algorithmic_TBps=round(S * bytes_per_pair / (ms.mean() * 1e-3) / 1e12, 3),
                     working_set_MB=round(S * 13.0, 1)))
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps(dict(points=n, launches_back_to_back=repeat, lib=os.environ.get("SICP_LIB", "product"), shapes=rows), indent=1))

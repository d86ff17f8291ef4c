"""Time the batched accumulate kernel alone (HIP events inside sicp_accumulate_batch).
usage: [ACC_MODE=gicp] bench_acc_batch.py [pairs] [points]   (ACC_MODE=gicp: the K = 1 kernel of SE3-GICP)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=n)
gicp = os.environ.get("ACC_MODE", "") == "gicp"
p = sicp.default_params(sicp.MODE_GICP if gicp else sicp.MODE_EM); p.num_classes = 0 if gicp else 11
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
es = []
for k in range(S):
    e = sicp.Engine(0, p)
    if not gicp: e.set_confusion(cm)
    e.set_source(ps, None if gicp else ls); e.set_target(pt, None if gicp else lt); e.correspondences(ident); es.append(e)
qts = np.tile(ident, (S, 1))
ms = []
for rep in range(8):
    out, t = sicp.accumulate_batch(es, qts, repeat=50)
    ms.append(t)
ms = np.array(ms[2:])
ref = es[0].accumulate(ident)
bytes_per_pair = 24 * n + 32 * (1 if gicp else 4) * n
print(f"pairs {S} points {n}: accumulate_batch {1e3 * ms.mean():.2f} us (min {1e3 * ms.min():.2f}) -> {S * bytes_per_pair / (ms.mean() * 1e-3) / 1e12:.2f} TB/s algorithmic; "
      f"equal to single: {np.array_equal(out[0], ref)}")

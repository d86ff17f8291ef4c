"""Walk statistics of the packet search (SICP_KNN_STATS): boxes tested and leaves scanned per packet.
usage: SICP_KNN_STATS=1 python tools/knn_stats.py [points]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from np_ref import mat_to_qt
os.environ.setdefault("SICP_KNN_STATS", "1")
sicp = importlib.import_module("semantic-icp_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ps, ls, pt, lt, T, cm = synth.lidar_pair(seed=2, n_points=n)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
with sicp.Engine(0, p) as e:
    e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt)
    print("--- K=4 at identity", flush=True); e.correspondences(np.array([0, 0, 0, 1, 0, 0, 0.0]))
    print("--- K=4 at the planted pose", flush=True); e.correspondences(mat_to_qt(T))
    print("--- k=20 self-search (source)", flush=True); e.covariances(sicp.SOURCE)

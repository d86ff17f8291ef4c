"""The search kernels as sicp_align_batch launches them: 16 different 100K-point pairs, one job per search,
8 jobs per launch (sicp_search_batch).  HIP-event time per search, per phase:
    k4_first   the K = 4 correspondence search from the curve position (first search of an align())
    k4_hinted  the same search seeded by the previous result (every later outer iteration)
    k20        the k = 20 self-search of the covariance neighbourhoods (source clouds)
    k4_weights k4_hinted as an EM-ICP align() launches it: the EM weights written by the search's epilogue (what = 3)
usage: bench_knn_jobs.py [phase|all] [pairs] [points] [reps]
Under rocprofv3 run ONE phase per invocation (tools/pmc_knn.py does); the last line says how many dispatches of
the phase's kernel belong to the timed repetitions."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from np_ref import mat_to_qt
phase = sys.argv[1] if len(sys.argv) > 1 else "all"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
sicp = importlib.import_module("semantic-icp_amd")
cm = synth.confusion_matrix(11)
# KNN_MODE=gicp: the K = 1 correspondence search of SE3-GICP (phases k4_* then time K = 1)
gicp = os.environ.get("KNN_MODE", "") == "gicp"
p = sicp.default_params(sicp.MODE_GICP if gicp else sicp.MODE_EM); p.num_classes = 0 if gicp else 11
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
es, poses = [], []
for k in range(S):
    seed = 2 + k
    motion = (1.0, 2.0) if seed == 2 else (0.5 + 0.11 * (seed % 11), -2.6 + 0.65 * (seed % 9))
    ps, ls, pt, lt, T, _ = synth.lidar_pair(seed=seed, n_points=n, motion=motion)
    e = sicp.Engine(0, p)
    if not gicp:
        e.set_confusion(cm)
    e.set_source(ps, None if gicp else ls); e.set_target(pt, None if gicp else lt)
    es.append(e)
    poses.append(mat_to_qt(T))   # the planted motion: where the later outer iterations search
poses = np.array(poses)
idents = np.tile(ident, (S, 1))
out = {}
launches_per_rep = (S + 7) // 8
def run(name, what, qts, hint, warm):
    for _ in range(warm):
        sicp.search_batch(es, qts, what=what, use_hint=hint, repeat=1)
    ms = [sicp.search_batch(es, qts, what=what, use_hint=hint, repeat=reps) for _ in range(3)]
    out[name] = {"us_per_search": 1e3 * min(ms) / S, "us_per_search_mean": 1e3 * float(np.mean(ms)) / S, "searches_per_rep": S,
                 "timed_dispatches": 3 * reps * launches_per_rep}
if phase in ("k4_first", "all"):
    run("k4_first", 0, idents, False, 1)
if phase in ("k4_hinted", "all"):
    sicp.search_batch(es, poses, what=0, use_hint=False, repeat=1)   # the search whose result seeds the next ones
    run("k4_hinted", 0, poses, True, 1)                             # late outer iterations: the pose barely moves
if phase in ("k20", "all"):
    run("k20", 1, None, False, 1)
if phase in ("k4_weights", "all") and not gicp:
    sicp.search_batch(es, poses, what=3, use_hint=False, repeat=1)
    run("k4_weights", 3, poses, True, 1)
for e in es:
    e.close()
print(json.dumps({"pairs": S, "points": n, "reps": reps, "phases": out}))

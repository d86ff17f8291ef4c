"""What can run BESIDE the accumulate kernel?  The accumulate launch over A pairs is kept resident for a while (the kernel
repeats its range SICP_ACC_INNER_REPEAT times inside one launch, as a fused multi-evaluation kernel would), and on a second
host thread / stream the search kernels of S other pairs are launched as sicp_align_batch launches them.  Wall time of each
alone and of both together: together ~ max(...) means the idle issue slots of one are filled by the other, together ~ sum
means they time-slice.  Run once per build / grid (SICP_LIB, SICP_ACC_GRID).
usage (GPU box): build a probe library first -- tools/build_variants.py probes=-DSICP_DEV_PROBES -- then\n  SICP_LIB=build_dbg/libsicp_probes.so SICP_ACC_INNER_REPEAT=40 corun_probe.py [acc_pairs] [search_pairs] [search_reps]\n(the product library has neither the in-kernel repeat loop nor the environment switch)"""
import importlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
sicp = importlib.import_module("semantic-icp_amd")
A = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 100000
inner = int(os.environ.get("SICP_ACC_INNER_REPEAT", "0"))
assert inner > 1, "set SICP_ACC_INNER_REPEAT"
pairs = [synth.lidar_pair(seed=2 + k, n_points=n) for k in range(4)]
cm = pairs[0][5]
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
def engines(count):
    out = []
    for k in range(count):
        ps, ls, pt, lt = pairs[k % 4][:4]
        e = sicp.Engine(0, p); e.set_confusion(cm); e.set_source(ps, ls); e.set_target(pt, lt); e.correspondences(ident); out.append(e)
    return out
acc_e, srch_e = engines(A), engines(S)
qa, qs = np.tile(ident, (A, 1)), np.tile(ident, (S, 1))
def run_acc(): sicp.accumulate_batch(acc_e, qa, repeat=1)
res = {}
for what, name in ((0, "k4_search"), (1, "k20_self_search")):
    def run_search(): sicp.search_batch(srch_e, qs, what=what, use_hint=False, repeat=reps)
    def wall(fns):
        th = [threading.Thread(target=f) for f in fns]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return 1e3 * (time.perf_counter() - t0)
    for f in (run_acc, run_search): f()   # warm
    t_acc = min(wall([run_acc]) for _ in range(3))
    t_s = min(wall([run_search]) for _ in range(3))
    t_both = min(wall([run_acc, run_search]) for _ in range(3))
    res[name] = dict(acc_alone_ms=round(t_acc, 2), search_alone_ms=round(t_s, 2), together_ms=round(t_both, 2),
                     sum_ms=round(t_acc + t_s, 2), max_ms=round(max(t_acc, t_s), 2),
                     overlap_gain=round((t_acc + t_s - t_both) / min(t_acc, t_s), 3))
print(json.dumps(dict(lib=os.environ.get("SICP_LIB", "product"), acc_grid=os.environ.get("SICP_ACC_GRID", "2 per CU"), acc_pairs=A, inner_repeat=inner,
                      search_pairs=S, search_reps=reps, result=res,
                      note="overlap_gain = (sum - together) / min(alone): 0 = time-sliced, 1 = the shorter one ran entirely inside the longer one")))

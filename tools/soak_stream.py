"""Soak of the registration stream under concurrency: several caller threads add clouds, submit registrations with random
options (fused labels, fresh features), release clouds early, poll -- while the library's worker runs the ticks.  Every
result is checked against a lone align() of that pair afterwards (bit equality), labels against sicp_fused_labels.
usage (GPU box): soak_stream.py [seconds] [threads] [points]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
scans = [synth.lidar_sequence_scan(31, i, n_points=n, n_az=500, step=(1.0 / 3.0, 2.0 / 3.0), period=40)[:2] for i in range(48)]
sicp = importlib.import_module("semantic-icp_amd")
cm = synth.confusion_matrix(11)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
results, lock = {}, threading.Lock()   # (source scan, target scan) -> list of (qt bytes, labels or None)
errors = []
def caller(tid, S, t_end):
    rng = np.random.default_rng(100 + tid)
    mine = {}      # ticket -> (s, t, wants labels)
    while time.time() < t_end:
        a, b = (int(v) for v in rng.choice(len(scans), 2, replace=False))
        ia, ib = S.add_cloud(*scans[a]), S.add_cloud(*scans[b])
        k = int(rng.integers(1, 4))
        for _ in range(k):
            fl, ff = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            mine[S.submit(ia, ib, ident, fused_labels=fl, fresh_features=ff)] = (a, b, fl)
        if rng.integers(0, 2):
            S.release_cloud(ia); S.release_cloud(ib)     # while its registrations are still in flight
            ia = ib = None
        for t, status, qt, st in S.poll(wait=int(rng.integers(0, 2))):
            if t not in mine:
                with lock: results.setdefault("foreign", []).append((t, status, qt, st))
                continue
            take(S, mine, t, status, qt)
        if ia is not None:
            S.release_cloud(ia); S.release_cloud(ib)
    return mine
def take(S, mine, t, status, qt):
    a, b, fl = mine.pop(t)
    if status != 0: errors.append(("status", status)); return
    lab = S.take_labels(t, len(scans[a][0])) if fl else None
    with lock: results.setdefault((a, b), []).append((qt.tobytes(), None if lab is None else lab.tobytes()))
t0 = time.time()
n_done = 0
with sicp.Stream(0, p, max_in_flight=24, confusion=cm) as S:
    left = [None] * n_threads
    def run(tid): left[tid] = caller(tid, S, t0 + secs)
    th = [threading.Thread(target=run, args=(k,)) for k in range(n_threads)]
    for t in th: t.start()
    for t in th: t.join()
    # results polled by one thread may belong to another: sort the rest out
    pending = {}
    for m in left: pending.update(m)
    for t, status, qt, st in results.pop("foreign", []) + S.drain():
        if t in pending: take(S, pending, t, status, qt)
    assert not pending, f"{len(pending)} registrations never came back"
    c = S.counters()
# check a sample against lone handles
pairs = list(results)
rng = np.random.default_rng(7)
bad = 0
for key in [pairs[i] for i in rng.choice(len(pairs), min(40, len(pairs)), replace=False)]:
    a, b = key
    with sicp.Engine(0, p) as e:
        e.set_confusion(cm); e.set_source(*scans[a]); e.set_target(*scans[b])
        q, st = e.align(ident)
        lab = e.fused_labels(q).tobytes()
    for qb, lb in results[key]:
        bad += qb != q.tobytes()
        bad += lb is not None and lb != lab
print(f"soak: {c['completed']} registrations in {time.time() - t0:.1f} s from {n_threads} threads, {len(pairs)} distinct pairs, errors {errors[:3]}, mismatches {bad}")
sys.exit(1 if (bad or errors) else 0)

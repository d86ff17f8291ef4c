"""Teardown soak: streams destroyed with registrations still queued / in flight / waiting for their fused labels, handles closed
right after uploads were queued, pools released in between -- nothing may crash, hang or leak device memory.
usage (GPU box): soak_teardown.py [cycles]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scans = [synth.lidar_sequence_scan(33, i, n_points=20000, n_az=700, step=(1.0 / 3.0, 2.0 / 3.0), period=40)[:2] for i in range(12)]
sicp = importlib.import_module("semantic-icp_amd")
cm = synth.confusion_matrix(11)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
rng = np.random.default_rng(1)
with sicp.Engine(0, p) as e:     # the reference result
    e.set_confusion(cm); e.set_source(*scans[1]); e.set_target(*scans[0])
    ref, _ = e.align(ident)
t0 = time.time()
peak = 0
for c in range(cycles):
    S = sicp.Stream(0, p, max_in_flight=int(rng.integers(1, 9)), confusion=cm)
    ids = [S.add_cloud(*sc) for sc in scans[: int(rng.integers(2, 12))]]
    tickets = [S.submit(ids[k + 1], ids[k], ident, fused_labels=bool(rng.integers(0, 2)), fresh_features=bool(rng.integers(0, 2))) for k in range(len(ids) - 1)]
    mode = c % 4
    if mode == 0:
        pass                                   # destroyed at once: everything still queued or in flight
    elif mode == 1:
        time.sleep(float(rng.uniform(0, 0.01)))  # destroyed somewhere in the middle
    elif mode == 2:
        got = S.poll(wait=1)                   # at least one result, then gone
        for t, status, qt, st in got:
            assert status == 0
            if t == tickets[0]: assert np.array_equal(qt, ref)
    else:
        res = {t: qt for t, status, qt, st in S.drain()}
        assert np.array_equal(res[tickets[0]], ref)
    S.close()
    if c % 10 == 9:
        h = sicp.Engine(0, p); h.set_confusion(cm); h.set_source(*scans[2]); h.set_target(*scans[3]); h.close()   # closed with its uploads just queued
        assert sicp.lib().sicp_release_pool(0) == 0
    peak = max(peak, sicp.memory_reserved(0))
assert sicp.lib().sicp_release_pool(0) == 0
left = sicp.memory_reserved(0)
print(f"teardown soak: {cycles} streams created and destroyed in {time.time() - t0:.1f} s, peak arena {peak / 2**20:.0f} MB, after release_pool {left / 2**20:.0f} MB")
sys.exit(0 if left == 0 else 1)

#!/usr/bin/env python3
"""Developer aid: the open-stream leg of bench.py alone (resident clouds), timed per phase.
usage (GPU box): tools/stream_probe.py [pairs] [points]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
def gen(k):
    return synth.lidar_sequence_scan(7, k, n_points=n, period=128)[:2]
import multiprocessing as mp
with mp.get_context("fork").Pool(min(64, os.cpu_count() or 8)) as pool:   # (before libsicp loads the HIP runtime)
    scans = pool.map(gen, range(n_pairs + 1))
sicp = importlib.import_module("semantic-icp_amd")
cm = synth.confusion_matrix(11)
p = sicp.default_params(sicp.MODE_EM); p.num_classes = 11; p.lm_batch = 4
variant = os.environ.get("PROBE_VARIANT", "")
if variant == "engine_solo":     # the stream never uses the persistent solve; a separate handle does, right before the clock starts
    p.lm_on_device = 2
    pe = sicp.default_params(sicp.MODE_EM); pe.num_classes = 11
    eng = sicp.Engine(0, pe); eng.set_confusion(cm); eng.set_source(*scans[1]); eng.set_target(*scans[0])
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
for rep in range(3):
    with sicp.Stream(0, p, max_in_flight=256, confusion=cm) as S:
        t = time.perf_counter()
        ids = [S.add_cloud(*sc) for sc in scans]
        t_add = time.perf_counter() - t
        t = time.perf_counter()
        S.submit(ids[-1], ids[0], ident)
        r0 = S.drain()
        t_first = time.perf_counter() - t
        if variant == "engine_solo":
            eng.align(ident)
        t0 = time.perf_counter()
        res = []
        marks = []
        for k in range(n_pairs):
            S.submit(ids[k + 1], ids[k], ident)
            if k % 64 == 0:
                res += S.poll(wait=0)
                marks.append((k, round((time.perf_counter() - t0) * 1e3), len(res)))
        t_sub = time.perf_counter() - t0
        res += S.drain()
        dt = time.perf_counter() - t0
        print("   submitted / ms / completed:", marks, flush=True)
        print(f"rep {rep}: add {t_add*1e3:.1f} ms, first registration + drain {t_first*1e3:.1f} ms (evals {r0[0][3]['total_evals']}, outer {r0[0][3]['outer_iters']}), "
              f"submit loop {t_sub*1e3:.1f} ms, total {dt*1e3:.1f} ms = {n_pairs/dt:.0f} pairs/s", flush=True)

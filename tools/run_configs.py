#!/usr/bin/env python3
"""Run the BASELINE.json configs 2-4 stand-ins through the engine and print one JSON line each
(timings from sicp_stats, pose error vs the planted transform).  Development aid."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from scipy.spatial.transform import Rotation
import synth
sicp = importlib.import_module("semantic-icp_amd")


def err(qt, T):
    x, y, z, w = qt[:4]
    R = Rotation.from_quat([x, y, z, w]).as_matrix()
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = qt[4:]
    D = np.linalg.inv(T) @ M
    return float(np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec())), float(np.linalg.norm(D[:3, 3]))


def run(name, mode, src, sl, tgt, tl, T, cm, C, reps=3, **kw):
    p = sicp.default_params(mode); p.num_classes = C
    for k, v in kw.items(): setattr(p, k, v)
    with sicp.Engine(0, p) as e:
        if cm is not None: e.set_confusion(cm)
        e.set_source(src, sl); e.set_target(tgt, tl); e.synchronize()      # first upload: allocations
        t0 = time.perf_counter(); e.set_source(src, sl); e.set_target(tgt, tl); e.synchronize(); t_set = time.perf_counter() - t0
        qt, st = e.align()
        t0 = time.perf_counter()
        for _ in range(reps): qt, st = e.align()
        dt = (time.perf_counter() - t0) / reps
    r, t = err(qt, T)
    print(json.dumps(dict(config=name, n_src=len(src), n_tgt=len(tgt), set_cloud_ms=round(1e3 * t_set, 1), align_ms=round(1e3 * dt, 2),
                          outer=st["outer_iters"], evals=st["total_evals"], corr_per_s=round(st["total_corr"] / dt),
                          rot_err=r, trans_err=t)), flush=True)


which = sys.argv[1:] or ["2", "2g", "3", "4"]
if "2" in which or "2g" in which:
    src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=2)
    if "2" in which: run("2: KITTI-like full scan, EM C=11", sicp.MODE_EM, src, sl, tgt, tl, T, cm, 11)
    if "2g" in which: run("2: KITTI-like full scan, SE3-GICP K=1", sicp.MODE_GICP, src, None, tgt, None, T, None, 0)
if "3" in which:
    src, sl, tgt, tl, T, cm = synth.rgbd_pair(seed=3)
    run("3: RGB-D frame pair, EM C=13 eps=1e-6", sicp.MODE_EM, src, sl, tgt, tl, T, cm, 13, epsilon=1e-6)
    run("3: RGB-D frame pair, SemanticICP", sicp.MODE_SEMANTIC, src, sl, tgt, tl, T, None, 0)
if "4" in which:
    src, sl, tgt, tl, T, cm = synth.facets_pair(seed=4)
    run("4: 1Mx1M facets, EM C=20", sicp.MODE_EM, src, sl, tgt, tl, T, cm, 20, reps=3)

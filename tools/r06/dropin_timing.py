#!/usr/bin/env python3
"""The drop-in path, timed: (1) every call of the reference drivers' per-pair sequences through the class
shims (tools/r06/dropin_calls.cc: exec/kitti_eval.cc:176-211 on a 100K x 100K LiDAR pair, exec/nyu_eval.cc:118-146 on a 307 200-point
RGB-D frame pair), (2) the reference's OWN kitti_eval / nyu_eval binaries (oracle/_ref/drivers, compiled unchanged) end to end on a
synthetic sequence, next to the headless drivers (one pair at a time, and -S: the open stream).
usage (GPU box): dropin_timing.py <out.json> [n_scans]"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
from test_host_shims import build_example, write_pcd
out_path = sys.argv[1]
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 13
tmp = tempfile.mkdtemp(prefix="dropin_")
from pathlib import Path
tp = Path(tmp)
HOST = os.path.join(ROOT, "semantic-icp_amd", "host")
exe = str(tp / "dropin_calls")
subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", HOST, os.path.join(ROOT, "tools", "r06", "dropin_calls.cc"),
                "-L", os.path.join(ROOT, "semantic-icp_amd"), "-lsicp", "-Wl,-rpath," + os.path.join(ROOT, "semantic-icp_amd"),
                "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe], check=True)
res = {"calls": [], "end_to_end": []}
# ---- (1) call by call
src, sl, tgt, tl, T, cm = synth.lidar_pair(seed=2, n_points=100000)
write_pcd(str(tp / "a.pcd"), src, sl, binary=True); write_pcd(str(tp / "b.pcd"), tgt, tl, binary=True)
np.savetxt(str(tp / "cm.txt"), cm, fmt="%.17g")
r = subprocess.run([exe, "kitti", str(tp / "a.pcd"), str(tp / "b.pcd"), str(tp / "cm.txt"), "8"], capture_output=True, text=True, timeout=600)
assert r.returncode == 0, r.stderr[-2000:]
res["calls"].append(json.loads(r.stdout.strip().splitlines()[-1]))
fr = synth.rgbd_pair(seed=3)
write_pcd(str(tp / "fa.pcd"), fr[0], fr[1], binary=True); write_pcd(str(tp / "fb.pcd"), fr[2], fr[3], binary=True)
r = subprocess.run([exe, "nyu", str(tp / "fa.pcd"), str(tp / "fb.pcd"), "-", "6"], capture_output=True, text=True, timeout=900)
assert r.returncode == 0, r.stderr[-2000:]
res["calls"].append(json.loads(r.stdout.strip().splitlines()[-1]))
json.dump(res, open(out_path, "w"), indent=1)
# ---- (2) end to end: a KITTI-like sequence of n_scans x 100K points, stride-3 pairs
scans, poses, cm = synth.lidar_sequence(seed=5, n_scans=n_scans, n_points=100000)
d = tp / "seq"; d.mkdir()
for k, (p, l) in enumerate(scans):
    write_pcd(str(d / f"{k:06d}.pcd"), p, l, binary=True)
gt = str(tp / "poses.txt"); np.savetxt(gt, poses[:, :3, :].reshape(n_scans, 12), fmt="%.17g")
cmf = str(tp / "cm.txt")
pairs = len(range(0, n_scans - 3, 3))
def timed(cmd, cwd):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=1800)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, (cmd, r.stderr[-2000:])
    return dt
ref = os.path.join(ROOT, "oracle", "_ref", "drivers", "kitti_eval")
if os.path.exists(ref):
    rd = tp / "ref_run"; rd.mkdir()
    dt = timed([ref, "-s", str(d), "-t", gt, "-m", cmf], str(rd))
    res["end_to_end"].append({"program": "oracle/_ref/drivers/kitti_eval (the reference's exec/kitti_eval.cc, unchanged, on the class shims)", "scans": n_scans, "points": 100000,
                              "pairs": pairs, "registrations": "EM-ICP<11> + SE3-GICP (+ the PCL-GICP stand-in, which returns its guess) per pair, 4 PCD loads per pair",
                              "wall_s": dt, "pairs_per_s": pairs / dt})
ours = build_example(tp, "kitti_eval_headless")
for label, extra in (("one pair at a time", []), ("-S 8: open streams, 8 registrations in flight", ["-S", "8"])):
    dt = timed([ours, "-s", str(d), "-t", gt, "-m", cmf, "-o", str(tp / ("o_" + str(len(extra)) + "_"))] + extra, str(tp))
    res["end_to_end"].append({"program": "kitti_eval_headless " + label, "scans": n_scans, "points": 100000, "pairs": pairs,
                              "registrations": "EM-ICP<11> + SE3-GICP per pair", "wall_s": dt, "pairs_per_s": pairs / dt})
json.dump(res, open(out_path, "w"), indent=1)
# ---- NYU-like: 3 frames of 307 200 points, the reference's nyu_eval against nyu_eval_headless
frames, fposes, fcm = synth.rgbd_sequence(seed=6, n_frames=4, stride=1)
fd = tp / "fseq"; fd.mkdir()
for k, (p, l) in enumerate(frames):
    write_pcd(str(fd / f"{k:04d}.pcd"), p, l, binary=True)
(tp / "pairs.txt").write_text("1 0\n2 1\n3 2\n")
ref = os.path.join(ROOT, "oracle", "_ref", "drivers", "nyu_eval")
if os.path.exists(ref):
    rd = tp / "nyu_ref"; rd.mkdir(); os.symlink(fd, rd / "fseq")
    dt = timed([ref, "-s", "fseq", "-t", str(tp / "pairs.txt")], str(rd))
    res["end_to_end"].append({"program": "oracle/_ref/drivers/nyu_eval (the reference's exec/nyu_eval.cc, unchanged, on the class shims)", "frames": 4, "points": len(frames[0][0]),
                              "pairs": 3, "registrations": "pcl_2_semantic x2 + SemanticICP + SE3-GICP + label metrics (host kd-tree) per pair", "wall_s": dt, "pairs_per_s": 3 / dt})
ours = build_example(tp, "nyu_eval_headless")
od = tp / "nyu_ours"; od.mkdir()
dt = timed([ours, "-s", str(fd), "-t", str(tp / "pairs.txt"), "-o", str(od / "o_"), "-c", "895"], str(tp))
res["end_to_end"].append({"program": "nyu_eval_headless", "frames": 4, "points": len(frames[0][0]), "pairs": 3, "wall_s": dt, "pairs_per_s": 3 / dt})
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))

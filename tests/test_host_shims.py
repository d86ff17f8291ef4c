"""The C++ class shims (semantic-icp_amd/host/) keep the reference's class/method names on top of
the C ABI.  CPU: the headless test_icp driver compiles against them and fails loudly without a
GPU.  GPU: it registers a PCD pair and matches the oracle."""
import importlib
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import oracle_lib as O
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "semantic-icp_amd", "host")
sicp = importlib.import_module("semantic-icp_amd")


def build_example(tmp_path, name="test_icp_headless"):
    sicp.build()
    exe = str(tmp_path / name)
    cmd = [
        "g++", "-std=c++17", "-O2", "-DEM_CLASSES=4", "-I", os.path.join(ROOT, "include"), "-I", HOST,
        os.path.join(HOST, "examples", name + ".cc"), "-L", os.path.join(ROOT, "semantic-icp_amd"), "-lsicp",
        "-Wl,-rpath," + os.path.join(ROOT, "semantic-icp_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe,
    ]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def write_pcd(path, xyz, labels, binary=False):
    n = len(xyz)
    hdr = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z label\nSIZE 4 4 4 4\nTYPE F F F U\n"
           f"COUNT 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {'binary' if binary else 'ascii'}\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("l", "<u4")])
            rec["x"], rec["y"], rec["z"], rec["l"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], labels
            f.write(rec.tobytes())
        else:
            for p, l in zip(xyz, labels):
                f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g} {int(l)}\n".encode())


def make_inputs(tmp_path, binary):
    src, sl, tgt, tl, T_gt = synth.config1_pair(seed=1, n_per_label=450)
    # add a class the driver drops (exec/test_icp.cc:53-55)
    extra = np.random.default_rng(0).uniform(0, 8, (60, 3)).astype(np.float32)
    src2, sl2 = np.concatenate([src, extra]), np.concatenate([sl, np.full(60, 3)]).astype(np.uint32)
    tgt2, tl2 = np.concatenate([tgt, extra + 1]), np.concatenate([tl, np.full(60, 3)]).astype(np.uint32)
    fs, ft, fm = str(tmp_path / "a.pcd"), str(tmp_path / "b.pcd"), str(tmp_path / "cm.txt")
    write_pcd(fs, src2, sl2, binary)
    write_pcd(ft, tgt2, tl2, binary)
    cm = synth.confusion_matrix(4)
    np.savetxt(fm, cm, fmt="%.17g")
    return (src2, sl2, tgt2, tl2, cm), (fs, ft, fm)


def test_example_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = build_example(tmp_path)
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    _, (fs, ft, fm) = make_inputs(tmp_path, binary=False)
    r = subprocess.run([exe, "-s", fs, "-t", ft, "-m", fm], capture_output=True, text=True)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr


def pose_delta(qa, qb):
    D = np.linalg.inv(O.se3_matrix(qa)) @ O.se3_matrix(qb)
    return np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()), np.linalg.norm(D[:3, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [False, True])
def test_headless_test_icp_matches_oracle(tmp_path, binary):
    exe = build_example(tmp_path)
    (src, sl, tgt, tl, cm), (fs, ft, fm) = make_inputs(tmp_path, binary)
    r = subprocess.run([exe, "-s", fs, "-t", ft, "-m", fm], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in r.stdout.splitlines():
        tok = line.split()
        if tok and tok[0] in ("SEMANTIC", "GICP", "EM"):
            got[tok[0]] = (np.array([float(v) for v in tok[1:8]]), int(tok[8]))
        if tok and tok[0] == "EM_FUSED":
            assert int(tok[1]) == len(src) and int(tok[2]) > 0.9 * len(src)
        if tok and tok[0] == "GICP_FINAL_CLOUD":
            assert int(tok[1]) == len(src)
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    keep_s, keep_t = sl != 3, tl != 3
    p = O.default_params(O.MODE_SEMANTIC)
    oq, _ = O.align(p, src[keep_s], sl[keep_s], tgt[keep_t], tl[keep_t], None, ident)
    rot, tr = pose_delta(got["SEMANTIC"][0], oq)
    assert rot < 1e-7 and tr < 1e-7
    p = O.default_params(O.MODE_GICP)
    oq, ost = O.align(p, src, None, tgt, None, None, ident)
    rot, tr = pose_delta(got["GICP"][0], oq)
    assert rot < 1e-7 and tr < 1e-7 and got["GICP"][1] == ost["outer_iters"]
    p = O.default_params(O.MODE_EM)
    p.num_classes = 4
    oq, ost = O.align(p, src, sl, tgt, tl, cm, ident)
    rot, tr = pose_delta(got["EM"][0], oq)
    assert rot < 1e-7 and tr < 1e-7 and got["EM"][1] == ost["outer_iters"]


# ------------------------------------------------------------------------------------------------
# f2: headless kitti_eval (exec/kitti_eval.cc:124-249 + exec/kitti_metrics.h)
# ------------------------------------------------------------------------------------------------
def make_sequence(tmp_path, n_scans=7, n_points=6000):
    scans, poses, cm = synth.lidar_sequence(seed=5, n_scans=n_scans, n_points=n_points)
    d = tmp_path / "seq"
    d.mkdir()
    for k, (p, l) in enumerate(scans):
        # one point beyond 40 m: the driver's range filter must drop it (exec/filter_range.h)
        p2 = np.concatenate([p, np.array([[45.0, 1.0, 0.0]], dtype=np.float32)])
        l2 = np.concatenate([l, np.array([1], dtype=np.uint32)])
        write_pcd(str(d / f"{k:06d}.pcd"), p2, l2, binary=True)
    gt = str(tmp_path / "poses.txt")
    np.savetxt(gt, poses[:, :3, :].reshape(n_scans, 12), fmt="%.17g")
    cmf = str(tmp_path / "cm.txt")
    np.savetxt(cmf, cm, fmt="%.17g")
    return scans, poses, cm, str(d), gt, cmf


def test_kitti_eval_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = build_example(tmp_path, "kitti_eval_headless")
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    _, _, _, d, gt, cmf = make_sequence(tmp_path, n_scans=4, n_points=500)
    r = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", str(tmp_path / "out_")], capture_output=True, text=True)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_kitti_eval_headless_rows_match_oracle(tmp_path):
    exe = build_example(tmp_path, "kitti_eval_headless")
    scans, poses, cm, d, gt, cmf = make_sequence(tmp_path, n_scans=10)
    prefix = str(tmp_path / "out_")
    r = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", prefix], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for fname, mode in (("EMICPkitti.csv", O.MODE_EM), ("se3GICPkitti.csv", O.MODE_GICP)):
        rows = [[float(v) for v in line.split(",")] for line in open(prefix + fname) if line.strip()]
        assert [int(r_[0]) for r_ in rows] == [0, 3, 6] and [int(r_[1]) for r_ in rows] == [3, 6, 9]  # stride-3 pairs
        for row in rows:
            a, b = int(row[0]), int(row[1])
            assert len(row) == 6 + 16 + 16 + 1
            T_est = np.array(row[22:38]).reshape(4, 4)
            dT = np.array(row[6:22]).reshape(4, 4)
            T_gt = np.linalg.inv(poses[a]) @ poses[b]
            assert np.allclose(dT, T_gt @ np.linalg.inv(T_est), atol=1e-9)
            rv = Rotation.from_matrix(dT[:3, :3]).as_rotvec()
            assert np.isclose(row[3], rv @ rv, rtol=1e-6, atol=1e-15) and np.isclose(row[4], dT[:3, 3] @ dT[:3, 3], rtol=1e-9)
            assert row[2] < 1e-4  # registers the pair: ||log(T_gt T^-1)||^2
            # same inputs through the oracle (EM sees the range-filtered cloud, GICP the raw file)
            p = O.default_params(mode)
            p.num_classes = 11
            (ps, ls), (pt, lt) = scans[b], scans[a]
            if mode == O.MODE_GICP:
                far = np.array([[45.0, 1.0, 0.0]], dtype=np.float32)
                ps, pt = np.concatenate([ps, far]), np.concatenate([pt, far])
            oq, ost = O.align(p, ps, ls if mode == O.MODE_EM else None, pt, lt if mode == O.MODE_EM else None,
                              cm if mode == O.MODE_EM else None, ident)
            D = np.linalg.inv(O.se3_matrix(oq)) @ T_est
            assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < 1e-7 and np.linalg.norm(D[:3, 3]) < 1e-7
            assert int(row[38]) == ost["outer_iters"]
    # -b 2: the same pairs registered two at a time in lock step (alignBatch / sicp_align_batch)
    prefix2 = str(tmp_path / "batch_")
    r2 = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", prefix2, "-b", "2"], capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr
    for fname in ("EMICPkitti.csv", "se3GICPkitti.csv"):
        one = [line.split(",") for line in open(prefix + fname) if line.strip()]
        two = [line.split(",") for line in open(prefix2 + fname) if line.strip()]
        assert len(one) == len(two)
        for ra, rb in zip(one, two):  # every column but the wall time (5) is identical text
            assert ra[:5] == rb[:5] and ra[6:] == rb[6:]
    # -r: every scan uploaded once and shared between its two registrations, features kept -- alone,
    # and in batches of 2 (the third pair's target is a cloud left on the GPU by the previous batch)
    for extra, tag in ((["-r"], "share1_"), (["-b", "2", "-r"], "share2_")):
        prefix3 = str(tmp_path / tag)
        r3 = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", prefix3] + extra, capture_output=True, text=True, timeout=900)
        assert r3.returncode == 0, r3.stderr
        for fname in ("EMICPkitti.csv", "se3GICPkitti.csv"):
            one = [line.split(",") for line in open(prefix + fname) if line.strip()]
            three = [line.split(",") for line in open(prefix3 + fname) if line.strip()]
            assert len(one) == len(three) == 3
            for ra, rb in zip(one, three):
                assert ra[:5] == rb[:5] and ra[6:] == rb[6:]


    # -S 4: the whole sequence as an open stream (sicp_stream_*), at most 4 registrations in flight
    prefix4 = str(tmp_path / "stream_")
    r4 = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", prefix4, "-S", "4"], capture_output=True, text=True, timeout=900)
    assert r4.returncode == 0, r4.stderr
    for fname in ("EMICPkitti.csv", "se3GICPkitti.csv"):
        one = [line.split(",") for line in open(prefix + fname) if line.strip()]
        four = [line.split(",") for line in open(prefix4 + fname) if line.strip()]
        assert len(one) == len(four) == 3
        for ra, rb in zip(one, four):
            assert ra[:5] == rb[:5] and ra[6:] == rb[6:]
    # -G 2: the pair list sharded into two contiguous runs, one host thread + one stream per method each, on devices
    # g % (devices visible) -- both on device 0 of a one-GPU box (the multi-GPU code path, oversubscribed); rows merged
    # in pair order: the same text again
    prefix5 = str(tmp_path / "sharded_")
    r5 = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", prefix5, "-G", "2", "-S", "4"], capture_output=True, text=True, timeout=900)
    assert r5.returncode == 0, r5.stderr
    assert "run 0: pairs [0, 1)" in r5.stdout and "run 1: pairs [1, 3)" in r5.stdout
    for fname in ("EMICPkitti.csv", "se3GICPkitti.csv"):
        one = [line.split(",") for line in open(prefix + fname) if line.strip()]
        five = [line.split(",") for line in open(prefix5 + fname) if line.strip()]
        assert len(one) == len(five) == 3
        for ra, rb in zip(one, five):
            assert ra[:5] == rb[:5] and ra[6:] == rb[6:]


@pytest.mark.gpu
def test_kitti_eval_headless_eight_runs_plumbing(tmp_path):
    """PLUMBING, NOT SCALING: -G 8 -- the pair list cut into eight contiguous runs, sixteen host threads and streams (one
    per run and method), devices g % (devices visible) -- on this box's one GPU: the rows, merged in pair order, are the
    text of a single run (exec/kitti_eval.cc:124-249 is the loop being sharded)."""
    exe = build_example(tmp_path, "kitti_eval_headless")
    scans, poses, cm, d, gt, cmf = make_sequence(tmp_path, n_scans=31, n_points=3000)   # 10 stride-3 pairs
    one, eight = str(tmp_path / "one_"), str(tmp_path / "eight_")
    r1 = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", one, "-G", "1", "-S", "2"], capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr
    r8 = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf, "-o", eight, "-G", "8", "-S", "2"], capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stderr
    runs = [l for l in r8.stdout.splitlines() if l.startswith("run ")]
    assert len(runs) == 8 and "run 0: pairs [0, 1)" in runs[0] and "run 7: pairs [8, 10)" in runs[7]
    for fname in ("EMICPkitti.csv", "se3GICPkitti.csv"):
        a = [line.split(",") for line in open(one + fname) if line.strip()]
        b = [line.split(",") for line in open(eight + fname) if line.strip()]
        assert len(a) == len(b) == 10
        for ra, rb in zip(a, b):
            assert ra[:5] == rb[:5] and ra[6:] == rb[6:]


# ------------------------------------------------------------------------------------------------
# f2: headless scenenet_eval (exec/scenenet_eval.cc:110-250 + exec/scenenet_metrics.h)
# ------------------------------------------------------------------------------------------------
def read_pcd_ascii(path):
    lines = open(path).read().splitlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("DATA"))
    a = np.array([[float(v) for v in l.split()] for l in lines[i + 1:] if l.strip()])
    return a[:, :3].astype(np.float32), a[:, 3].astype(np.uint32)


@pytest.mark.gpu
def test_scenenet_eval_headless_rows_and_fused_labels(tmp_path):
    exe = build_example(tmp_path, "scenenet_eval_headless")
    frames, poses, cm = synth.rgbd_sequence(seed=6, n_frames=3)
    d = tmp_path / "seq"
    d.mkdir()
    for k, (p, l) in enumerate(frames):
        write_pcd(str(d / f"{k:04d}.pcd"), p, l, binary=False)
    gt = str(tmp_path / "gt.txt")
    with open(gt, "w") as f:  # SceneNet rows: inverse pose (4x4 row major) + frame index
        for k, P in enumerate(poses):
            f.write(" ".join(f"{v:.17g}" for v in np.linalg.inv(P).reshape(-1)) + f" {k}\n")
    cmf = str(tmp_path / "cm.txt")
    np.savetxt(cmf, cm, fmt="%.17g")
    prefix = str(tmp_path / "out_")
    r = subprocess.run([exe, "-s", str(d), "-t", gt, "-m", cmf, "-o", prefix], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for fname, mode in (("EMICPscenenet.csv", O.MODE_EM), ("se3GICPscenenet.csv", O.MODE_GICP)):
        rows = [[float(v) for v in line.split(",")] for line in open(prefix + fname) if line.strip()]
        assert [(int(r_[0]), int(r_[1])) for r_ in rows] == [(0, 1), (1, 2)]  # consecutive frames
        for row in rows:
            a, b = int(row[0]), int(row[1])
            T_est = np.array(row[22:38]).reshape(4, 4)
            T_gt = np.linalg.inv(poses[a]) @ poses[b]
            assert np.allclose(np.array(row[6:22]).reshape(4, 4), T_gt @ np.linalg.inv(T_est), atol=1e-9)
            p = O.default_params(mode)
            p.num_classes = 13
            p.epsilon = 1e-6
            (ps, ls), (pt, lt) = frames[b], frames[a]
            oq, ost = O.align(p, ps, ls if mode == O.MODE_EM else None, pt, lt if mode == O.MODE_EM else None,
                              cm if mode == O.MODE_EM else None, ident)
            D = np.linalg.inv(O.se3_matrix(oq)) @ T_est
            assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < 1e-4 and np.linalg.norm(D[:3, 3]) < 1e-3
            assert int(row[38]) == ost["outer_iters"]
            if mode == O.MODE_EM:  # fused labels written as <prefix><source index>.pcd
                fp, fl = read_pcd_ascii(f"{prefix}{b}.pcd")
                assert np.allclose(fp, ps, atol=1e-5)
                olab = O.fused_labels(p, ps, ls, pt, lt, cm, oq)
                assert np.array_equal(fl, olab)
    # -b 2: both frame pairs registered together (alignBatch): identical rows and label files
    prefix2 = str(tmp_path / "b_")
    r2 = subprocess.run([exe, "-s", str(d), "-t", gt, "-m", cmf, "-o", prefix2, "-b", "2"], capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr
    for fname in ("EMICPscenenet.csv", "se3GICPscenenet.csv"):
        one = [line.split(",") for line in open(prefix + fname) if line.strip()]
        two = [line.split(",") for line in open(prefix2 + fname) if line.strip()]
        assert len(one) == len(two) and all(ra[:5] == rb[:5] and ra[6:] == rb[6:] for ra, rb in zip(one, two))
    for k in (1, 2):
        assert open(f"{prefix}{k}.pcd").read() == open(f"{prefix2}{k}.pcd").read()
    # -S 2: the sequence as an open stream per method; the fused labels come back with each registration
    # (SICP_SUBMIT_FUSED_LABELS), every frame uploaded once: identical rows and label files
    prefix3 = str(tmp_path / "s_")
    r3 = subprocess.run([exe, "-s", str(d), "-t", gt, "-m", cmf, "-o", prefix3, "-S", "2"], capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stderr
    for fname in ("EMICPscenenet.csv", "se3GICPscenenet.csv"):
        one = [line.split(",") for line in open(prefix + fname) if line.strip()]
        three = [line.split(",") for line in open(prefix3 + fname) if line.strip()]
        assert len(one) == len(three) and all(ra[:5] == rb[:5] and ra[6:] == rb[6:] for ra, rb in zip(one, three))
    for k in (1, 2):
        assert open(f"{prefix}{k}.pcd").read() == open(f"{prefix3}{k}.pcd").read()


# ------------------------------------------------------------------------------------------------
# f2: headless nyu_eval (exec/nyu_eval.cc:45-222 + exec/nyu_metrics.h)
# ------------------------------------------------------------------------------------------------
def test_nyu_eval_compiles_and_fails_loudly_without_gpu(tmp_path):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    exe = build_example(tmp_path, "nyu_eval_headless")
    frames, poses, cm = synth.rgbd_sequence(seed=6, n_frames=2, stride=16)
    d = tmp_path / "seq"
    d.mkdir()
    for k, (p, l) in enumerate(frames):
        write_pcd(str(d / f"{k:04d}.pcd"), p, l)
    (tmp_path / "pairs.txt").write_text("0 1\n")
    r = subprocess.run([exe, "-s", str(d), "-t", str(tmp_path / "pairs.txt"), "-o", str(tmp_path / "o_")], capture_output=True, text=True)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_nyu_eval_headless_poses_and_label_agreement(tmp_path):
    from scipy.spatial import cKDTree

    exe = build_example(tmp_path, "nyu_eval_headless")
    frames, poses, cm = synth.rgbd_sequence(seed=6, n_frames=3, stride=4)
    d = tmp_path / "seq"
    d.mkdir()
    for k, (p, l) in enumerate(frames):
        write_pcd(str(d / f"{k:04d}.pcd"), p, l, binary=True)
    (tmp_path / "pairs.txt").write_text("1 0\n2 1 0\n")  # rows of indices; consecutive entries are (source, target)
    prefix = str(tmp_path / "o_")
    r = subprocess.run([exe, "-s", str(d), "-t", str(tmp_path / "pairs.txt"), "-o", prefix, "-c", "16"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("pair ")]
    assert [(l[1], l[2]) for l in lines] == [("1->0", "SICP"), ("1->0", "se3GICP"), ("2->1", "SICP"), ("2->1", "se3GICP"),
                                             ("1->0", "SICP"), ("1->0", "se3GICP")]
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    summary = {name: [[float(v) for v in row.split(",")] for row in open(prefix + name) if row.strip()]
               for name in ("SICPnyu.csv", "se3GICPnyu.csv")}
    counters = {"SICP": 0, "se3GICP": 0}
    for l in lines:
        s_i, t_i = (int(v) for v in l[1].split("->"))
        which = l[2]
        qt = np.array([float(v) for v in l[4:11]])
        (ps, ls), (pt, lt) = frames[s_i], frames[t_i]
        p = O.default_params(O.MODE_SEMANTIC)
        zero = np.zeros(len(ps), np.uint32), np.zeros(len(pt), np.uint32)
        oq, _ = O.align(p, ps, ls if which == "SICP" else zero[0], pt, lt if which == "SICP" else zero[1], None, ident)
        D = np.linalg.inv(O.se3_matrix(oq)) @ O.se3_matrix(qt)
        assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < 1e-4 and np.linalg.norm(D[:3, 3]) < 1e-3
        # the metric (exec/nyu_metrics.h:36-84) recomputed from the driver's pose
        M = O.se3_matrix(qt).astype(np.float32)
        moved = (ps @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
        dist, nn = cKDTree(pt).query(moved)
        keep = dist.astype(np.float32) ** 2 < 25.0
        ratio, mean_d, total = np.mean(ls[keep] == lt[nn[keep]]), dist[keep].mean(), keep.sum()
        row = summary[which + "nyu.csv"][counters[which]]
        counters[which] += 1
        assert int(row[0]) == s_i and abs(row[3] - total) <= 2
        assert abs(row[1] - ratio) < 2e-3 and abs(row[2] - mean_d) < 1e-4 and abs(float(l[12]) - row[1]) < 1e-5
    # per-pair label files and the accumulated confusion matrix
    lab = np.loadtxt(f"{tmp_path}/Label1-o_SICPnyu.csv", delimiter=",")
    assert lab.shape[1] == 2 and len(lab) == int(summary["SICPnyu.csv"][-1][3])
    mat = np.loadtxt(f"{tmp_path}/Matrixo_SICPnyu.csv")
    assert mat.shape == (16, 16) and mat.sum() == sum(r_[3] for r_ in summary["SICPnyu.csv"])
    # -S 3: every pair of the test file through two open streams (labelled / single class), a frame uploaded once per
    # stream however many pairs name it: the same lines in the same order.  (The class shim hands the engine the points
    # grouped by label, the stream takes them in file order: two points in one cell of the curve may swap places on the
    # device, so poses are compared to 1e-9 rather than as text.)
    prefix_s = str(tmp_path / "s_")
    rs = subprocess.run([exe, "-s", str(d), "-t", str(tmp_path / "pairs.txt"), "-o", prefix_s, "-c", "16", "-S", "3"], capture_output=True,
                        text=True, timeout=900)
    assert rs.returncode == 0, rs.stderr
    lines_s = [l.split() for l in rs.stdout.splitlines() if l.startswith("pair ")]
    assert [(l[1], l[2]) for l in lines_s] == [(l[1], l[2]) for l in lines]
    for a, b in zip(lines, lines_s):
        qa, qb = np.array([float(v) for v in a[4:11]]), np.array([float(v) for v in b[4:11]])
        D = np.linalg.inv(O.se3_matrix(qa)) @ O.se3_matrix(qb)
        assert np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec()) < 1e-9 and np.linalg.norm(D[:3, 3]) < 1e-9
        assert abs(float(a[12]) - float(b[12])) < 1e-4
    for name in ("SICPnyu.csv", "se3GICPnyu.csv"):
        rows_s = [[float(v) for v in row.split(",")] for row in open(prefix_s + name) if row.strip()]
        assert len(rows_s) == len(summary[name])
        for ra, rb in zip(summary[name], rows_s):
            assert ra[0] == rb[0] and abs(ra[3] - rb[3]) <= 2 and abs(ra[1] - rb[1]) < 2e-3


@pytest.mark.gpu
def test_semantic_point_cloud_lives_on_the_device_and_handles_are_pooled(tmp_path):
    """tests/cpp/semantic_cloud_check.cc: the SemanticPointCloud of the class shims owns one device-resident cloud that
    SemanticIterativeClosestPoint::align shares (no flatten / upload / covariance pass per align), labeledCovariances is
    fetched on first read and equals what addSemanticCloud computes per label cloud (also when read after transform()),
    align() equals the flat C-ABI path bit for bit, and sicp_destroy / sicp_create recycle handles."""
    sicp.build()
    src, sl, tgt, tl, T_gt = synth.config1_pair(seed=1, n_per_label=700)
    write_pcd(str(tmp_path / "a.pcd"), src, sl, binary=True)
    write_pcd(str(tmp_path / "b.pcd"), tgt, tl, binary=True)
    exe = str(tmp_path / "semantic_cloud_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", HOST, os.path.join(ROOT, "tests", "cpp", "semantic_cloud_check.cc"),
                    "-L", os.path.join(ROOT, "semantic-icp_amd"), "-lsicp", "-Wl,-rpath," + os.path.join(ROOT, "semantic-icp_amd"),
                    "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe, str(tmp_path / "a.pcd"), str(tmp_path / "b.pcd")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {line.split()[0]: int(line.split()[1]) for line in r.stdout.splitlines() if line.strip()}
    assert got == {"lazy_covariances_equal_per_label_clouds": 1, "covariances_after_transform_are_those_of_the_added_cloud": 1,
                   "align_on_shared_device_clouds_equals_flat_c_abi": 1, "destroyed_handle_is_reused_and_fresh": 1,
                   "supplied_covariances_of_the_engines_form_are_taken": 1, "supplied_covariances_that_are_no_covariances_are_refused_loudly": 1,
                   "supplied_covariances_of_general_form_are_taken": 1,
                   "release_pool_frees_parked_handles": 1}, r.stdout

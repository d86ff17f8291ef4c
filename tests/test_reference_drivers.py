"""The reference's OWN driver programs -- exec/kitti_eval.cc, nyu_eval.cc, scenenet_eval.cc, roc_eval.cc, test_icp.cc and the
utilities make_semantic.cc, pcd_read.cc, pcd_write.cc -- compiled UNCHANGED, where they lie under /root/reference, against
the class shims in semantic-icp_amd/host and LINKED with libsicp.so (`link unchanged`, BASELINE.json north_star; SURVEY.md
section 8b).  Recipe: oracle/build_ref_drivers.py, outputs in oracle/_ref/drivers/ (they travel to the GPU box; no reference
source is copied).

CPU (build container only, where the reference tree exists): every driver compiles and links with the reference's own
flags; the two that need no GPU run; one that needs a GPU fails loudly without it.
GPU: the reference's main() programs run on the MI355X engine, and what they write -- the CSV rows of kitti_metrics.h /
scenenet_metrics.h / nyu_metrics.h, the fused-label PCD files -- agrees with this repository's headless drivers on the
same inputs (to the 6 significant digits the reference prints)."""
import glob
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import pcd_files
import synth
from test_host_shims import build_example, make_sequence, read_pcd_ascii, write_pcd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref_drivers as ref  # noqa: E402

sicp = importlib.import_module("semantic-icp_amd")
needs_reference = pytest.mark.skipif(not ref.available(), reason="the reference tree only exists in the build container")


def driver(name):
    """The travelled binary (GPU box) or a fresh build (build container)."""
    path = os.path.join(ref.OUT, name)
    if ref.available():
        sicp.build()
        ref.build([name])
    if not os.path.exists(path):
        pytest.skip(f"oracle/_ref/drivers/{name} was not built (no reference tree when the snapshot was taken)")
    return path


@needs_reference
def test_every_reference_driver_compiles_and_links_unchanged(tmp_path):
    sicp.build()
    built = ref.build(out_dir=str(tmp_path))          # raises with the compiler output on any error
    assert sorted(built) == sorted(ref.DRIVERS)
    for name, path in built.items():
        assert os.access(path, os.X_OK)
        nm = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        uses_engine = "sicp_" in nm
        # the registration drivers bind to the C ABI; the two PCD utilities are header-only
        assert uses_engine == (name not in ("pcd_read", "pcd_write")), (name, nm)
    # the reference's flags really are the ones used (CMakeLists.txt:5)
    assert "-std=c++11" in ref.command("kitti_eval", "x") and "-O3" in ref.command("kitti_eval", "x")


@needs_reference
def test_reference_pcd_utilities_run_on_the_host(tmp_path):
    ref.build(["pcd_write", "pcd_read"], out_dir=str(tmp_path))
    r = subprocess.run([str(tmp_path / "pcd_write")], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Saved 5 data points" in r.stderr
    pts, lab = read_pcd_ascii(str(tmp_path / "test_pcd.pcd"))
    assert pts.shape == (5, 3) and np.all(np.abs(pts) < 1024) and len(lab) == 5   # (`1024 * rand()` overflows int in the reference)
    # exec/pcd_read.cc reads ./cloudA.pcd and lists the points whose label is not 0 -- here from a binary_compressed file
    xyz = np.arange(30, dtype=np.float32).reshape(10, 3) / 4
    labels = np.array([0, 3, 0, 0, 7, 0, 1, 0, 0, 12], dtype=np.uint32)
    pcd_files.write_pcd(str(tmp_path / "cloudA.pcd"), xyz, labels, "binary_compressed")
    r = subprocess.run([str(tmp_path / "pcd_read")], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rows = [[float(v) for v in line.split()] for line in r.stdout.splitlines() if line.startswith("    ")]
    want = [[*xyz[i], labels[i]] for i in range(10) if labels[i]]
    assert np.allclose(rows, want) and "Loaded 10 data points" in r.stdout


@needs_reference
def test_reference_kitti_eval_fails_loudly_without_gpu(tmp_path):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    exe = driver("kitti_eval")
    _, _, _, d, gt, cmf = make_sequence(tmp_path, n_scans=4, n_points=300)
    r = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no usable HIP device" in r.stderr   # an exception out of the class shim: no silent CPU path


def rows_of(path):
    return [[float(v) for v in line.split(",")] for line in open(path) if line.strip()]


def one(pattern):
    found = glob.glob(pattern)
    assert len(found) == 1, (pattern, found)
    return found[0]


def same_rows(ref_rows, our_rows, time_col=5):
    """The reference prints 6 significant digits; the headless drivers print 17."""
    assert len(ref_rows) == len(our_rows) and len(ref_rows) > 0
    for a, b in zip(ref_rows, our_rows):
        assert len(a) == len(b)
        for c, (x, y) in enumerate(zip(a, b)):
            if c == time_col:
                continue
            assert abs(x - y) <= 2e-5 * max(abs(x), abs(y)) + 2e-9, (c, x, y)


@pytest.mark.gpu
def test_reference_kitti_eval_runs_on_the_engine(tmp_path):
    exe = driver("kitti_eval")
    scans, poses, cm, d, gt, cmf = make_sequence(tmp_path, n_scans=10)
    run_dir = tmp_path / "ref_run"
    run_dir.mkdir()
    r = subprocess.run([exe, "-s", d, "-t", gt, "-m", cmf], cwd=run_dir, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "SICP FINAL MSE" in r.stdout and "se3GICP FINAL MSE" in r.stdout
    ours = build_example(tmp_path, "kitti_eval_headless")
    prefix = str(tmp_path / "ours_")
    r2 = subprocess.run([ours, "-s", d, "-t", gt, "-m", cmf, "-o", prefix], capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr
    for name in ("EMICPkitti.csv", "se3GICPkitti.csv"):
        ref_rows = rows_of(one(str(run_dir / ("*" + name))))
        assert [(int(x[0]), int(x[1])) for x in ref_rows] == [(0, 3), (3, 6), (6, 9)]
        same_rows(ref_rows, rows_of(prefix + name))
        assert all(row[2] < 1e-4 for row in ref_rows)   # the reference's own error column: the pairs are registered
    # the bootstrap column is the identity guess (= the ground-truth motion as its error); PCL's GICP column is the stand-in,
    # which registers nothing and says so in a way no reader of the file can miss: NaN in every derived number
    init = rows_of(one(str(run_dir / "*initkitti.csv")))
    pcl_files = [f for f in glob.glob(str(run_dir / "*GICPkitti.csv")) if not f.endswith("se3GICPkitti.csv")]
    assert len(pcl_files) == 1
    pclg = rows_of(pcl_files[0])
    for a, b in zip(init, pclg):
        T_gt = np.linalg.inv(poses[int(a[0])]) @ poses[int(a[1])]
        assert np.allclose(np.array(a[6:22]).reshape(4, 4), T_gt, atol=1e-4)
        assert np.all(np.isnan(b[2:5]))


@pytest.mark.gpu
def test_reference_scenenet_eval_runs_on_the_engine(tmp_path):
    exe = driver("scenenet_eval")
    frames, poses, cm = synth.rgbd_sequence(seed=6, n_frames=3)
    d = tmp_path / "seq"
    d.mkdir()
    for k, (p, l) in enumerate(frames):
        pcd_files.write_pcd(str(d / f"{k:04d}.pcd"), p, l, "binary_compressed" if k == 1 else "ascii")
    gt = str(tmp_path / "gt.txt")
    with open(gt, "w") as f:
        for k, P in enumerate(poses):
            f.write(" ".join(f"{v:.17g}" for v in np.linalg.inv(P).reshape(-1)) + f" {k}\n")
    cmf = str(tmp_path / "cm.txt")
    np.savetxt(cmf, cm, fmt="%.17g")
    run_dir = tmp_path / "ref_run"
    run_dir.mkdir()
    r = subprocess.run([exe, "-s", str(d), "-t", gt, "-m", cmf], cwd=run_dir, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ours = build_example(tmp_path, "scenenet_eval_headless")
    prefix = str(tmp_path / "ours_")
    r2 = subprocess.run([ours, "-s", str(d), "-t", gt, "-m", cmf, "-o", prefix], capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr
    for name in ("EMICPscenenet.csv", "se3GICPscenenet.csv"):
        same_rows(rows_of(one(str(run_dir / ("*" + name)))), rows_of(prefix + name))
    for k in (1, 2):   # emicp.getFusedLabels -> savePCDFileASCII("<source index>.pcd"), exec/scenenet_eval.cc:193-198
        assert open(run_dir / f"{k}.pcd").read() == open(f"{prefix}{k}.pcd").read()


@pytest.mark.gpu
def test_reference_nyu_eval_runs_on_the_engine(tmp_path):
    exe = driver("nyu_eval")
    frames, poses, cm = synth.rgbd_sequence(seed=6, n_frames=3, stride=4)
    d = tmp_path / "seq"
    d.mkdir()
    for k, (p, l) in enumerate(frames):
        write_pcd(str(d / f"{k:04d}.pcd"), p, l, binary=True)
    (tmp_path / "pairs.txt").write_text("1 0\n2 1 0\n")
    run_dir = tmp_path / "ref_run"
    run_dir.mkdir()
    # relative paths: exec/nyu_eval.cc:103-107 takes the cloud number from ALL digits of the path it was given
    os.symlink(d, run_dir / "seq")
    r = subprocess.run([exe, "-s", "seq", "-t", str(tmp_path / "pairs.txt")], cwd=run_dir, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    ours = build_example(tmp_path, "nyu_eval_headless")
    our_dir = tmp_path / "our_run"
    our_dir.mkdir()
    r2 = subprocess.run([ours, "-s", str(d), "-t", str(tmp_path / "pairs.txt"), "-o", str(our_dir / "o_"), "-c", "895"], capture_output=True,
                        text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr
    # summary rows `cloud number, label agreement, mean NN distance, pairs within 5 m` (exec/nyu_metrics.h:77-81): the
    # reference's host KdTreeFLANN search against the headless driver's GPU search
    # (column 0, the "cloud number": exec/nyu_eval.cc:103-107 moves the path's digits to the front with std::remove_if and
    #  never erases the tail, so std::stoi reads "0001" + the untouched rest "0001.pcd" of "seq/0001.pcd": 10001 for frame 1)
    def ref_cloud_number(path):
        digits = "".join(ch for ch in path if ch.isdigit())
        moved = digits + path[len(digits):]
        lead = ""
        for ch in moved:
            if not ch.isdigit():
                break
            lead += ch
        return int(lead)
    for name in ("SICPnyu.csv", "se3GICPnyu.csv"):
        ref_rows, our_rows = rows_of(one(str(run_dir / ("[0-9]*" + name)))), rows_of(str(our_dir / ("o_" + name)))
        assert [int(r_[0]) for r_ in ref_rows] == [ref_cloud_number(f"seq/{int(r_[0]):04d}.pcd") for r_ in our_rows]
        same_rows([r_[1:] for r_ in ref_rows], [r_[1:] for r_ in our_rows], time_col=-1)
    # the accumulated 895 x 895 confusion matrix, printed by Eigen's operator<< there and by the headless driver here
    ref_mat = np.loadtxt(one(str(run_dir / "Matrix*SICPnyu.csv")))
    our_mat = np.loadtxt(str(our_dir / "Matrixo_SICPnyu.csv"))
    assert ref_mat.shape == (895, 895) and np.array_equal(ref_mat, our_mat)
    # per-evaluation label pairs
    ref_lab = np.loadtxt(one(str(run_dir / "Label10001-*SICPnyu.csv")), delimiter=",")
    our_lab = np.loadtxt(str(our_dir / "Label1-o_SICPnyu.csv"), delimiter=",")
    assert np.array_equal(ref_lab, our_lab)


@pytest.mark.gpu
def test_reference_test_icp_and_roc_eval_run_to_the_end(tmp_path):
    exe = driver("test_icp")
    src, sl, tgt, tl, T_gt = synth.config1_pair(seed=1, n_per_label=450)
    write_pcd(str(tmp_path / "a.pcd"), src, sl, binary=True)
    write_pcd(str(tmp_path / "b.pcd"), tgt, tl, binary=False)
    r = subprocess.run([exe, "-s", str(tmp_path / "a.pcd"), "-t", str(tmp_path / "b.pcd")], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Time Multiclass" in r.stdout and "Time Single Class" in r.stdout and "GICP transform" in r.stdout
    # exec/make_semantic.cc: reads ./cloudB.pcd, zeroes the labels, splits it with pcl_2_semantic (covariances on the GPU) and
    # prints the covariances that are NaN -- none for a cloud without degenerate neighbourhoods
    ms = driver("make_semantic")
    ms_dir = tmp_path / "ms"
    ms_dir.mkdir()
    pcd_files.write_pcd(str(ms_dir / "cloudB.pcd"), tgt, tl, "binary_compressed")
    r = subprocess.run([ms], cwd=ms_dir, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Loaded {len(tgt)} data points" in r.stdout and "Labels:" in r.stdout and "nan" not in r.stdout.lower()
    # exec/roc_eval.cc: label pairs of registered clouds against separately labelled ground-truth clouds
    roc = driver("roc_eval")
    for sub in ("pred", "gt"):
        (tmp_path / sub).mkdir()
        write_pcd(str(tmp_path / sub / "0000.pcd"), tgt, tl, binary=True)
        write_pcd(str(tmp_path / sub / "0001.pcd"), src, sl, binary=True)
    run_dir = tmp_path / "roc_run"
    run_dir.mkdir()
    # (exec/roc_eval.cc:170-176 moves a cloud by PCL-GICP's matrix and looks its points up in a kd-tree: with the stand-in's
    #  all-NaN matrix its own metrics code indexes out of range, so this run asks the stand-in for the finite guess)
    r = subprocess.run([roc, "-s", str(tmp_path / "pred"), "-t", str(tmp_path / "gt")], cwd=run_dir, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SICP_PCL_GICP_RETURNS_GUESS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    pairs = np.loadtxt(one(str(run_dir / "*SICProc.csv")), delimiter=",")
    assert len(pairs) == len(src) and np.mean(pairs[:, 0] == pairs[:, 1]) > 0.9   # registered: nearest neighbours share labels
    T = np.loadtxt(one(str(run_dir / "*SICPtransform.csv")))
    assert T.shape == (4, 4) and np.allclose(T, T_gt, atol=5e-2)

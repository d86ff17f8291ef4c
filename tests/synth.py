"""Seeded synthetic stand-ins for the BASELINE.json configs.

No datasets are available offline, so every config is generated procedurally
with numpy's PCG64 (deterministic for a fixed numpy version, identical here and
on the GPU box).  All clouds are float32 xyz + uint32 labels in 1..C.
"""
from __future__ import annotations

import numpy as np


def confusion_matrix(C: int, diag: float = 0.8) -> np.ndarray:
    """Row-stochastic CxC confusion matrix: `diag` on the diagonal, rest uniform
    (the kind of file exec/read_confusion_matrix.h:6-19 loads)."""
    cm = np.full((C, C), (1.0 - diag) / (C - 1))
    np.fill_diagonal(cm, diag)
    return cm


def pose_matrix(rot_deg: float, axis, trans) -> np.ndarray:
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    th = np.deg2rad(rot_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = trans
    return M


def _noisy_labels(rng, labels, cm, frac):
    """Re-draw `frac` of the labels from the confusion-matrix row of the true class."""
    labels = labels.copy()
    n = labels.shape[0]
    flip = rng.random(n) < frac
    C = cm.shape[0]
    for i in np.nonzero(flip)[0]:
        row = cm[labels[i] - 1].copy()
        row[labels[i] - 1] = 0
        row /= row.sum()
        labels[i] = 1 + rng.choice(C, p=row)
    return labels


# -----------------------------------------------------------------------------
# config 1: test_icp.cc-style pair, 3 labels x 700 points
# -----------------------------------------------------------------------------
def config1_pair(seed: int = 1, n_per_label: int = 700, sigma: float = 0.01):
    """Three noisy patches (two planes + one curved sheet) in a 10 m box, labels
    {1, 2, 4} (3/10/11 are dropped by exec/test_icp.cc:53-55).  Target = the same
    surfaces re-sampled, moved by T_gt (3 deg about (1,2,3), (0.2,-0.1,0.05) m)."""
    rng = np.random.default_rng(seed)
    T_gt = pose_matrix(3.0, (1, 2, 3), (0.2, -0.1, 0.05))

    def sample(n):
        pts, lab = [], []
        u, v = rng.uniform(0, 6, n), rng.uniform(0, 6, n)          # floor patch z = 0.1x
        pts.append(np.stack([u + 1, v + 1, 0.1 * u], 1)); lab.append(np.full(n, 1))
        u, v = rng.uniform(0, 6, n), rng.uniform(0, 3, n)          # wall patch x = 8 - 0.2y
        pts.append(np.stack([8 - 0.2 * u, u + 1, v], 1)); lab.append(np.full(n, 2))
        u, v = rng.uniform(0, 5, n), rng.uniform(0, 3, n)          # curved sheet
        pts.append(np.stack([u + 2, 9 - 0.15 * (u - 2.5) ** 2, v + 0.5], 1)); lab.append(np.full(n, 4))
        p = np.concatenate(pts) + rng.normal(0, sigma, (3 * n, 3))
        return p, np.concatenate(lab).astype(np.uint32)

    ps, ls = sample(n_per_label)
    pt, lt = sample(n_per_label)
    pt = pt @ T_gt[:3, :3].T + T_gt[:3, 3]
    # interleave so that label order of first appearance is not sorted
    o = rng.permutation(ps.shape[0]); ps, ls = ps[o], ls[o]
    o = rng.permutation(pt.shape[0]); pt, lt = pt[o], lt[o]
    return ps.astype(np.float32), ls, pt.astype(np.float32), lt, T_gt


# -----------------------------------------------------------------------------
# config 2 / metric point: 64-ring LiDAR ray-cast of a procedural street
# -----------------------------------------------------------------------------
def _street(rng):
    boxes = []  # (lo(3), hi(3), label)
    # parked cars (label 3), vans (4), along both kerbs
    for side in (-1, 1):
        x = -55.0
        while x < 55:
            x += rng.uniform(6, 14)
            L, W, H = rng.uniform(3.8, 5.2), rng.uniform(1.6, 2.0), rng.uniform(1.3, 2.2)
            y0 = side * rng.uniform(4.0, 5.0)
            lab = 3 if H < 1.8 else 4
            boxes.append((np.array([x, y0 - W / 2, 0]), np.array([x + L, y0 + W / 2, H]), lab))
    # building bays (labels 5..8) protruding from the walls
    for side in (-1, 1):
        x = -60.0
        while x < 60:
            w = rng.uniform(5, 12)
            d = rng.uniform(0.3, 2.0)
            lab = int(rng.integers(5, 9))
            lo = np.array([x, 9.0 - d if side > 0 else -9.0, 0])
            hi = np.array([x + w, 9.0 if side > 0 else -9.0 + d, rng.uniform(4, 9)])
            boxes.append((lo, hi, lab))
            x += w + rng.uniform(1, 5)
    poles = []  # (cx, cy, r, h, label): poles 9, trunks 10, signs 11
    for _ in range(40):
        poles.append((rng.uniform(-55, 55), rng.choice([-1, 1]) * rng.uniform(5.5, 7.5), rng.uniform(0.08, 0.3),
                      rng.uniform(2.5, 7), int(rng.integers(9, 12))))
    return boxes, poles


def _raycast(origin, dirs, boxes, poles):
    """Nearest hit of each ray against ground z=0 (label 1), walls y=+-9 (label 2),
    axis-aligned boxes and vertical cylinders.  Returns (t, label)."""
    n = dirs.shape[0]
    tbest = np.full(n, np.inf)
    lab = np.zeros(n, dtype=np.uint32)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (0.0 - origin[2]) / dirs[:, 2]
        ok = (t > 0.5) & np.isfinite(t)
        upd = ok & (t < tbest); tbest[upd] = t[upd]; lab[upd] = 1
        for ysign in (-1.0, 1.0):
            t = (ysign * 9.0 - origin[1]) / dirs[:, 1]
            z = origin[2] + t * dirs[:, 2]
            ok = (t > 0.5) & np.isfinite(t) & (z >= 0) & (z <= 12)
            upd = ok & (t < tbest); tbest[upd] = t[upd]; lab[upd] = 2
        inv = 1.0 / dirs
        for lo, hi, bl in boxes:
            t0 = (lo - origin) * inv
            t1 = (hi - origin) * inv
            tn = np.minimum(t0, t1).max(axis=1)
            tf = np.maximum(t0, t1).min(axis=1)
            ok = (tn <= tf) & (tn > 0.5)
            upd = ok & (tn < tbest); tbest[upd] = tn[upd]; lab[upd] = bl
        for cx, cy, r, h, pl in poles:
            ox, oy = origin[0] - cx, origin[1] - cy
            a = dirs[:, 0] ** 2 + dirs[:, 1] ** 2
            b = 2 * (ox * dirs[:, 0] + oy * dirs[:, 1])
            c = ox * ox + oy * oy - r * r
            disc = b * b - 4 * a * c
            t = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
            z = origin[2] + t * dirs[:, 2]
            ok = np.isfinite(t) & (t > 0.5) & (z >= 0) & (z <= h)
            upd = ok & (t < tbest); tbest[upd] = t[upd]; lab[upd] = pl
    return tbest, lab


def _lidar_scan(rng, sensor_pose, boxes, poles, n_az, max_range, sigma):
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    az = np.deg2rad(np.arange(n_az) * (360.0 / n_az) + rng.uniform(0, 360.0 / n_az))
    e, a = np.meshgrid(elev, az, indexing="ij")
    d_s = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], -1).reshape(-1, 3)
    R, o = sensor_pose[:3, :3], sensor_pose[:3, 3]
    t, lab = _raycast(o, d_s @ R.T, boxes, poles)
    keep = np.isfinite(t) & (t < max_range)
    t = t[keep] + rng.normal(0, sigma, keep.sum())
    pts = d_s[keep] * t[:, None]            # sensor frame
    return pts, lab[keep]


def lidar_pair(seed: int = 2, n_points: int | None = None, C: int = 11, n_az: int = 2250,
               max_range: float = 40.0, sigma: float = 0.01, label_noise: float = 0.10,
               motion=(1.0, 2.0)):
    """KITTI-like pair (config 2; `n_points=100_000` gives the metric point).
    Ego-motion: `motion[0]` m forward + `motion[1]` deg yaw between the scans.
    Returns src, src_labels, tgt, tgt_labels, T_gt (src frame -> tgt frame), cm."""
    rng = np.random.default_rng(seed)
    boxes, poles = _street(rng)
    cm = confusion_matrix(C)
    pose_t = np.eye(4); pose_t[:3, 3] = (0.0, 0.3, 1.73)
    step = pose_matrix(motion[1], (0, 0, 1), (motion[0], 0.0, 0.0))
    pose_s = pose_t @ step
    out = []
    for pose in (pose_s, pose_t):
        p, l = _lidar_scan(rng, pose, boxes, poles, n_az, max_range, sigma)
        l = _noisy_labels(rng, l, cm, label_noise)
        if n_points is not None:
            if p.shape[0] < n_points:
                raise ValueError(f"scan has only {p.shape[0]} points; raise n_az")
            sel = np.sort(rng.choice(p.shape[0], n_points, replace=False))
            p, l = p[sel], l[sel]
        out += [p.astype(np.float32), l.astype(np.uint32)]
    T_gt = np.linalg.inv(pose_t) @ pose_s
    return out[0], out[1], out[2], out[3], T_gt, cm


def lidar_sequence(seed: int = 5, n_scans: int = 7, n_points: int | None = 20000, C: int = 11, n_az: int = 2250,
                   max_range: float = 40.0, sigma: float = 0.01, label_noise: float = 0.10, step=(1.0 / 3.0, 2.0 / 3.0)):
    """KITTI-odometry-like sequence (config 5 stand-in): the sensor of config 2 moves `step[0]` m
    forward and yaws `step[1]` deg per scan.  Returns [(xyz, labels)], world poses (n,4,4), cm.
    The reference's driver registers scan n+3 (source) onto scan n (target), exec/kitti_eval.cc:124-129."""
    rng = np.random.default_rng(seed)
    boxes, poles = _street(rng)
    cm = confusion_matrix(C)
    pose = np.eye(4); pose[:3, 3] = (-3.0, 0.3, 1.73)
    inc = pose_matrix(step[1], (0, 0, 1), (step[0], 0.0, 0.0))
    scans, poses = [], []
    for _ in range(n_scans):
        p, l = _lidar_scan(rng, pose, boxes, poles, n_az, max_range, sigma)
        l = _noisy_labels(rng, l, cm, label_noise)
        if n_points is not None:
            sel = np.sort(rng.choice(p.shape[0], n_points, replace=False))
            p, l = p[sel], l[sel]
        scans.append((p.astype(np.float32), l.astype(np.uint32)))
        poses.append(pose.copy())
        pose = pose @ inc
    return scans, np.stack(poses), cm


def lidar_sequence_scan(seed: int, i: int, n_points: int | None = 100_000, C: int = 11, n_az: int = 2250, max_range: float = 40.0,
                        sigma: float = 0.01, label_noise: float = 0.10, step=(1.0, 2.0), wobble: float = 0.15, period: int | None = None):
    """Scan `i` of a long KITTI-odometry-like sequence, generated independently of the others (so a
    sequence can be produced by a process pool): the street of `seed`, the sensor of config 2 after
    `i` steps of `step[0]` m forward and `step[1]` deg yaw, modulated per scan by +-`wobble` so that
    consecutive registrations differ in difficulty.  The heading swings between -8 and +8 degrees with a
    period of 16 scans, which keeps the vehicle inside the 120 m street.  With `period` the vehicle drives
    `period` steps up the street, the same steps back (in reverse gear), and so on -- a sequence of any
    length stays inside the street; scans at the same position differ in their noise.
    Returns (xyz, labels, world pose 4x4)."""
    boxes, poles = _street(np.random.default_rng(seed))
    cm = confusion_matrix(C)
    pose = np.eye(4); pose[:3, 3] = (-40.0, 0.3, 1.73)
    n_steps = i if period is None else period - abs(i % (2 * period) - period)
    for j in range(n_steps):
        r = np.random.default_rng([seed, 7919, j])
        f = 1.0 + wobble * r.uniform(-1, 1)
        yaw = step[1] * (1.0 if (j % 16) < 4 or (j % 16) >= 12 else -1.0)
        pose = pose @ pose_matrix(yaw, (0, 0, 1), (0.6 * step[0] * f, 0.0, 0.0))
    rng = np.random.default_rng([seed, i])
    p, l = _lidar_scan(rng, pose, boxes, poles, n_az, max_range, sigma)
    l = _noisy_labels(rng, l, cm, label_noise)
    if n_points is not None:
        sel = np.sort(rng.choice(p.shape[0], n_points, replace=False))
        p, l = p[sel], l[sel]
    return p.astype(np.float32), l.astype(np.uint32), pose


# -----------------------------------------------------------------------------
# config 3: RGB-D frame pair of a box room (pinhole depth render)
# -----------------------------------------------------------------------------
def _rgbd_scene(rng, width, height, stride):
    room_lo, room_hi = np.array([-3.0, -2.5, 0.0]), np.array([3.0, 2.5, 2.8])
    cuboids = []
    for k in range(9):
        c = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-2.0, 2.0), 0.0])
        s = np.array([rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.5)])
        cuboids.append((c - [s[0] / 2, s[1] / 2, 0], c + [s[0] / 2, s[1] / 2, s[2]], 5 + k))
    f = 0.9 * width
    us, vs = np.meshgrid(np.arange(0, width, stride) + 0.5, np.arange(0, height, stride) + 0.5)
    d_c = np.stack([(us - width / 2) / f, (vs - height / 2) / f, np.ones_like(us)], -1).reshape(-1, 3)
    d_c /= np.linalg.norm(d_c, axis=1, keepdims=True)
    # camera looks along +x of the room, z up: camera axes (x right, y down, z fwd)
    Rwc = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    pose0 = np.eye(4); pose0[:3, :3] = Rwc; pose0[:3, 3] = (-2.6, 0.1, 1.4)
    return room_lo, room_hi, cuboids, d_c, pose0


def _rgbd_render(rng, pose, scene):
    room_lo, room_hi, cuboids, d_c, _ = scene
    R, o = pose[:3, :3], pose[:3, 3]
    dw = d_c @ R.T
    n = dw.shape[0]
    lab = np.zeros(n, dtype=np.uint32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dw
        # inside of the room box: exit distance, label by face (floor 1, ceiling 2, walls 3/4)
        t0 = (room_lo - o) * inv; t1 = (room_hi - o) * inv
        tf3 = np.maximum(t0, t1)
        face = np.argmin(tf3, axis=1)
        tbest = tf3.min(axis=1)
        up = dw[np.arange(n), 2] > 0
        lab[:] = np.where(face == 2, np.where(up, 2, 1), np.where(face == 0, 3, 4))
        for lo, hi, bl in cuboids:
            t0 = (lo - o) * inv; t1 = (hi - o) * inv
            tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
            ok = (tn <= tf) & (tn > 0.3)
            upd = ok & (tn < tbest); tbest[upd] = tn[upd]; lab[upd] = bl
    z = tbest * d_c[:, 2]
    z = z + rng.normal(0, 1.0, n) * 0.0012 * z * z      # depth noise ~ z^2
    pts = d_c * (z / d_c[:, 2])[:, None]
    keep = np.isfinite(z) & (z > 0.3) & (z < 8.0)
    return pts[keep], lab[keep]


def rgbd_pair(seed: int = 3, width: int = 640, height: int = 480, C: int = 13,
              label_noise: float = 0.10, stride: int = 1):
    rng = np.random.default_rng(seed)
    cm = confusion_matrix(C)
    scene = _rgbd_scene(rng, width, height, stride)
    pose_t = scene[4]
    step = pose_matrix(3.0, (0.2, 1.0, 0.1), (0.03, 0.01, 0.04))
    pose_s = pose_t @ step
    out = []
    for pose in (pose_s, pose_t):
        p, l = _rgbd_render(rng, pose, scene)
        l = _noisy_labels(rng, l, cm, label_noise)
        out += [p.astype(np.float32), l.astype(np.uint32)]
    T_gt = np.linalg.inv(pose_t) @ pose_s
    return out[0], out[1], out[2], out[3], T_gt, cm


def rgbd_sequence(seed: int = 6, n_frames: int = 3, width: int = 640, height: int = 480, C: int = 13,
                  label_noise: float = 0.10, stride: int = 4):
    """SceneNet-like sequence: the camera of config 3 moves by a small step per frame.  Returns
    [(xyz, labels)], camera-to-world poses (n,4,4), cm.  exec/scenenet_eval.cc registers frame n+1
    (source) onto frame n (target)."""
    rng = np.random.default_rng(seed)
    cm = confusion_matrix(C)
    scene = _rgbd_scene(rng, width, height, stride)
    pose = scene[4]
    step = pose_matrix(1.5, (0.2, 1.0, 0.1), (0.02, 0.01, 0.03))
    frames, poses = [], []
    for _ in range(n_frames):
        p, l = _rgbd_render(rng, pose, scene)
        l = _noisy_labels(rng, l, cm, label_noise)
        frames.append((p.astype(np.float32), l.astype(np.uint32)))
        poses.append(pose.copy())
        pose = pose @ step
    return frames, np.stack(poses), cm


# -----------------------------------------------------------------------------
# config 4: N points on random planar facets in a cube
# -----------------------------------------------------------------------------
def facets_pair(seed: int = 4, n_points: int = 1_000_000, n_facets: int = 200, C: int = 20,
                cube: float = 100.0, sigma: float = 0.01, label_noise: float = 0.10):
    rng = np.random.default_rng(seed)
    cm = confusion_matrix(C)
    centers = rng.uniform(-cube / 2, cube / 2, (n_facets, 3))
    normals = rng.normal(size=(n_facets, 3)); normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    a = np.cross(normals, rng.normal(size=(n_facets, 3))); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(normals, a)
    size = rng.uniform(4, 12, n_facets)
    flabel = (1 + np.arange(n_facets) % C).astype(np.uint32)
    T_gt = pose_matrix(2.0, (0.3, -0.2, 1.0), (0.35, -0.25, 0.25))

    def sample():
        f = rng.integers(0, n_facets, n_points)
        u = rng.uniform(-1, 1, n_points) * size[f]
        v = rng.uniform(-1, 1, n_points) * size[f]
        p = centers[f] + u[:, None] * a[f] + v[:, None] * b[f] + rng.normal(0, sigma, (n_points, 3))
        lab = flabel[f].copy()
        flip = rng.random(n_points) < label_noise
        lab[flip] = rng.integers(1, C + 1, flip.sum()).astype(np.uint32)
        return p, lab

    ps, ls = sample()
    pt, lt = sample()
    pt = pt @ T_gt[:3, :3].T + T_gt[:3, 3]
    return ps.astype(np.float32), ls, pt.astype(np.float32), lt, T_gt, cm

"""Synthetic edge-alignment workloads (numpy + scipy only) shared by tests/ and bench.py.

The reference bundles five TUM RGB-D grabs and nothing else; the larger configurations of
BASELINE.json (C2 twin, C3 pyramid, C4 batch, C5 roofline stress) are synthesised here the way
SURVEY.md §8(d) prescribes: an exact Euclidean distance transform of random line segments, edge
points sampled on those segments, random depths, back-projected and moved by a planted pose so
that the planted pose is the minimiser.  Seeds are the configuration numbers.
"""
import numpy as np


def quat_from_axis_angle(axis, angle_rad):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle_rad / 2)], np.sin(angle_rad / 2) * axis])


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def rotation_angle_between(q1, q2):
    """angle (rad) of the relative rotation between two unit quaternions"""
    q1 = np.asarray(q1, float) / np.linalg.norm(q1)
    q2 = np.asarray(q2, float) / np.linalg.norm(q2)
    d = abs(float(np.dot(q1, q2)))
    return 2.0 * np.arccos(min(1.0, d))


def random_segments(H, W, n_segments, rng, min_len=40.0, max_len=None):
    max_len = max_len or 0.35 * min(H, W)
    min_len = min(min_len, 0.5 * max_len)
    p0 = np.stack([rng.uniform(8, W - 9, n_segments), rng.uniform(8, H - 9, n_segments)], axis=1)
    ang = rng.uniform(0, 2 * np.pi, n_segments)
    ln = rng.uniform(min_len, max_len, n_segments)
    p1 = p0 + np.stack([np.cos(ang), np.sin(ang)], axis=1) * ln[:, None]
    p1[:, 0] = np.clip(p1[:, 0], 8, W - 9)
    p1[:, 1] = np.clip(p1[:, 1], 8, H - 9)
    return p0, p1


def rasterize(H, W, p0, p1):
    mask = np.zeros((H, W), dtype=bool)
    for a, b in zip(p0, p1):
        n = int(np.ceil(max(abs(b[0] - a[0]), abs(b[1] - a[1])))) + 1
        s = np.linspace(0.0, 1.0, n)
        u = np.rint(a[0] + s * (b[0] - a[0])).astype(int)
        v = np.rint(a[1] + s * (b[1] - a[1])).astype(int)
        mask[v, u] = True
    return mask


def exact_edt(mask):
    from scipy import ndimage
    return ndimage.distance_transform_edt(~mask).astype(np.float32)


def make_problem(H, W, n_points, n_segments, seed, fx, fy, cx, cy,
                 planted_q=(1.0, 0.0, 0.0, 0.0), planted_t=(0.0, 0.0, 0.0),
                 depth_range=(0.5, 5.0), normalize=False, order="raster", pixel_centres=True):
    """Returns dict(image HxW float32, grid WxH float64 (Grid2D view), xyz (n,3) float64 in frame A,
    q_true, t_true, K).  b_T_a = (q_true, t_true) maps xyz onto edge pixels of `image`."""
    rng = np.random.default_rng(seed)
    p0, p1 = random_segments(H, W, n_segments, rng)
    mask = rasterize(H, W, p0, p1)
    dt = exact_edt(mask)
    if normalize:
        dt = (dt / dt.max()).astype(np.float32)
    vv, uu = np.nonzero(mask)
    if pixel_centres:
        sel = rng.integers(0, vv.size, n_points) if n_points > vv.size else rng.choice(vv.size, n_points, replace=False)
        u = uu[sel].astype(np.float64)
        v = vv[sel].astype(np.float64)
    else:
        seg = rng.integers(0, n_segments, n_points)
        s = rng.random(n_points)
        u = p0[seg, 0] + s * (p1[seg, 0] - p0[seg, 0])
        v = p0[seg, 1] + s * (p1[seg, 1] - p0[seg, 1])
    if order == "raster":
        idx = np.lexsort((u, np.floor(v)))
    elif order == "random":
        idx = rng.permutation(n_points)
    else:
        raise ValueError(order)
    u, v = u[idx], v[idx]
    Z = rng.uniform(depth_range[0], depth_range[1], n_points)
    b = np.stack([(u - cx) * Z / fx, (v - cy) * Z / fy, Z], axis=1)  # frame B
    q = np.asarray(planted_q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    t = np.asarray(planted_t, dtype=np.float64)
    R = quat_to_R(q)
    a = (b - t) @ R  # a = R^T (b - t)
    return dict(image=dt, grid=np.ascontiguousarray(dt.astype(np.float64).T), xyz=a,
                q_true=q, t_true=t, K=(fx, fy, cx, cy), mask=mask)


def config_c2_twin(seed=2, n_points=50000):
    """C2: single 640x480 frame pair, ~5e4 edge points, fp64 (synthetic twin of the bundled pair)."""
    q = quat_from_axis_angle([0.3, -1.0, 0.5], np.deg2rad(0.8))
    return make_problem(480, 640, n_points, 420, seed, 525.0, 525.0, 319.5, 239.5,
                        planted_q=q, planted_t=(0.004, -0.003, 0.01), normalize=True)


def config_c5(seed=5, n_points=1000000, order="raster"):
    """C5: 1e6-point cloud into a 2048x1536 un-normalised exact EDT, planted pose
    (0.5 deg about (1,2,3)/sqrt(14), t = (1,-0.5,2) cm)."""
    q = quat_from_axis_angle([1.0, 2.0, 3.0], np.deg2rad(0.5))
    return make_problem(1536, 2048, n_points, 4000, seed, 1680.0, 1680.0, 1023.5, 767.5,
                        planted_q=q, planted_t=(0.01, -0.005, 0.02), normalize=False, order=order)


def config_c3_levels(seed=3):
    """C3: 3-level pyramid, 1280x960 / 640x480 / 320x240 with 140k / 45k / 15k points; the same
    planted pose on every level, intrinsics scaled by 2^-l with the half-pixel rule
    c_l = (c_0 + 0.5) / 2^l - 0.5."""
    q = quat_from_axis_angle([0.2, 1.0, -0.4], np.deg2rad(1.0))
    t = (0.01, 0.004, -0.015)
    fx0, fy0, cx0, cy0 = 1050.0, 1050.0, 639.5, 479.5
    levels = []
    for l, (H, W, n, segs) in enumerate([(960, 1280, 140000, 1500), (480, 640, 45000, 420), (240, 320, 15000, 120)]):
        s = 2.0 ** l
        levels.append(make_problem(H, W, n, segs, seed * 10 + l, fx0 / s, fy0 / s,
                                   (cx0 + 0.5) / s - 0.5, (cy0 + 0.5) / s - 0.5,
                                   planted_q=q, planted_t=t, normalize=False))
    return levels


def rigid_4x4(q, t):
    T = np.eye(4)
    T[:3, :3] = quat_to_R(np.asarray(q, float) / np.linalg.norm(q))
    T[:3, 3] = t
    return T


def make_stereo_problem(H, W, n1, n2, seed, K1, K2, T12, planted_q, planted_t, distortion=None, normalize=True):
    """Two residual families on one pose, the shape of the reference's stereo tests
    (standalone_edge_align.cpp:791-803): camera-1 points with EAResidue[Ex], camera-2 points with
    EAResidueSecondCam[Ex] whose pose is T12 * b_T_a * T12^-1.  Points are planted on the edges of each
    camera's DT image (through the distortion model when given, by fixed-point inversion)."""
    Tp = rigid_4x4(planted_q, planted_t)
    T2 = T12 @ Tp @ np.linalg.inv(T12)
    out = []
    for cam, (K, n, Tcam) in enumerate(((K1, n1, Tp), (K2, n2, T2))):
        pr = make_problem(H, W, n, max(8, n // 100), seed * 10 + cam, *K, normalize=normalize)
        fx, fy, cx, cy = K
        # make_problem planted identity: xyz are frame-B points on the edges; undistort, then move back by Tcam
        b = pr["xyz"]
        if distortion is not None:
            k1, k2, p1, p2, k3 = distortion
            xd, yd = b[:, 0] / b[:, 2], b[:, 1] / b[:, 2]
            x, y = xd.copy(), yd.copy()
            for _ in range(50):
                r2 = x * x + y * y
                D = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
                x = (xd - 2 * p1 * x * y - p2 * (r2 + 2 * x * x)) / D
                y = (yd - 2 * p2 * x * y - p1 * (r2 + 2 * y * y)) / D
            b = np.stack([x * b[:, 2], y * b[:, 2], b[:, 2]], axis=1)
        a = (b - Tcam[:3, 3]) @ Tcam[:3, :3]   # a = R^T (b - t)
        out.append(dict(xyz=a, grid=pr["grid"], image=pr["image"], K=K))
    return out


# ---- BASELINE config C4: 256 frame pairs, 32 per GPU ----------------------------------------------------------------
# TUM fr1_desk is not available offline: the 256 problems are the 20 ordered pairs of the five bundled grabs
# (standalone/rgb-d, copied to tests/golden/rgbd as fixtures) x 12-13 seeded initial-pose perturbations (rotation <= 1
# degree, translation <= 2 cm, seed 4).  Problem i = pair (i mod 20), perturbation (i div 20); perturbation 0 is the
# identity start of the reference's drivers.

C4_TOTAL = 256
TUM_K = (525.0, 525.0, 319.5, 239.5)  # standalone_edge_align.cpp:151-160


def load_bundled_frames(directory):
    """{1..5: (bgr uint8 HxWx3 as cv::imread returns it, depth uint16 HxW)} from tests/golden/rgbd (PIL)."""
    import os
    from PIL import Image
    out = {}
    for k in range(1, 6):
        rgb = np.asarray(Image.open(os.path.join(directory, "rgb_%d.png" % k)).convert("RGB"), dtype=np.uint8)
        dep = np.asarray(Image.open(os.path.join(directory, "depth_%d.png" % k))).astype(np.uint16)
        out[k] = (np.ascontiguousarray(rgb[:, :, ::-1]), np.ascontiguousarray(dep))
    return out


def config_c4_specs(total=C4_TOTAL, seed=4):
    """[(ref frame, now frame, q0 (4,), t0 (3,))] for problems 0..total-1, identical on every rank."""
    pairs = [(a, b) for a in range(1, 6) for b in range(1, 6) if a != b]
    rng = np.random.default_rng(seed)
    n_pert = (total + len(pairs) - 1) // len(pairs)
    perts = [(np.array([1.0, 0, 0, 0]), np.zeros(3))]
    for _ in range(1, n_pert):
        axis = rng.normal(size=3)
        q = quat_from_axis_angle(axis, np.deg2rad(rng.uniform(0.0, 1.0)))
        d = rng.normal(size=3)
        perts.append((q, d / np.linalg.norm(d) * rng.uniform(0.0, 0.02)))
    return [(pairs[i % len(pairs)][0], pairs[i % len(pairs)][1], perts[i // len(pairs)][0].copy(), perts[i // len(pairs)][1].copy())
            for i in range(total)]

"""Randomised soak, third part: the frame producers of the three flavours (Laplacian / Canny / ROS) on random images of
random sizes with random thresholds, masks, flags and half-resolution steps -- every stage the debug hooks expose, bit
for bit against oracle/preprocess_np.py.  usage: python scripts/soak_preprocess.py [seconds] [seed]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi
from oracle import preprocess_np as pp
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
K = (525.0, 525.0, 319.5, 239.5)
t_end = time.time() + budget
n_cases = n_fail = 0
counts = {}


def image(H, W):
    kind = int(rng.integers(6))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    if kind == 0:
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == 1:    # blobs + steps + noise
        img = np.zeros((H, W, 3))
        for c in range(3):
            a = 60 * np.sin(xx / (5 + 3 * c) + rng.random() * 6) + 60 * np.cos(yy / (7 + c) + rng.random() * 6)
            a += 80 * ((xx - W * rng.random()) ** 2 + (yy - H * rng.random()) ** 2 < (15 + 10 * c) ** 2)
            a += 50 * (xx > W * rng.random()) + rng.normal(0, float(rng.uniform(0, 12)), (H, W))
            img[:, :, c] = a
        img -= img.min()
        return (img * (255.0 / max(img.max(), 1.0))).astype(np.uint8)
    if kind == 2:    # a few rectangles on a flat ground: long straight chains, big flat regions
        img = np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            r0, c0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            img[r0:r0 + int(rng.integers(1, H)), c0:c0 + int(rng.integers(1, W))] = rng.integers(0, 256, 3)
        return img
    if kind == 3:    # smooth ramp (gradients just around the thresholds) + sparse salt
        g = (xx * rng.uniform(0, 3) + yy * rng.uniform(0, 3)) % 256
        img = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
        img[rng.random((H, W)) < 0.01] = 255
        return img
    if kind == 4:    # checkerboard of random pitch: every pixel a candidate
        p = int(rng.integers(1, 9))
        g = ((((xx // p) + (yy // p)) % 2) * int(rng.integers(20, 256))).astype(np.uint8)
        return np.repeat(g[:, :, None], 3, axis=2)
    img = np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)   # one lit pixel / one line
    if rng.random() < 0.5:
        img[int(rng.integers(0, H)), int(rng.integers(0, W))] = 255 - img[0, 0]
    else:
        img[int(rng.integers(0, H)), :] = 255 - img[0, 0]
    return img


def check(name, ok, info):
    global n_fail
    counts[name] = counts.get(name, 0) + 1
    if not ok:
        n_fail += 1
        print("FAIL", name, info, flush=True)


while time.time() < t_end:
    H, W = int(rng.integers(3, 220)), int(rng.integers(3, 320))
    if rng.random() < 0.15:
        H, W = [(3, 3), (3, 257), (257, 3), (16, 16), (17, 64), (64, 65), (128, 256)][int(rng.integers(7))]
    bgr = image(H, W)
    depth = rng.integers(0, 30000, (H, W)).astype(np.uint16)
    depth[rng.random((H, W)) < rng.uniform(0, 0.5)] = 0
    info = (n_cases, H, W)
    P = capi.Problem(*K, dtype=capi.EA_F64 if rng.random() < 0.5 else capi.EA_F32)
    # -- Laplacian flavour
    thr = int(rng.integers(0, 120)); med = bool(rng.random() < 0.5); nrm = bool(rng.random() < 0.5)
    lap = pp.edge_strength(bgr)
    Bm = np.where(lap > thr, 0, 255).astype(np.uint8)
    mask = pp.median_blur3_u8(Bm) if med else Bm
    st = P.set_now_frame(bgr, threshold=thr, median=med, normalize=nrm, debug=True)
    check("lap", np.array_equal(st["lap"], lap) and np.array_equal(st["mask"], mask), info)
    if (mask == 0).any():
        check("lap.chamfer", np.array_equal(st["chamfer"].astype(np.int64), pp.chamfer3x3_fixed(mask == 0)), info)
        dist = pp.distance_transform_l2_3(mask)
        check("lap.dt", np.array_equal(st["dt"], pp.normalize_minmax_f32(dist) if nrm else dist), info + (thr, med, nrm))
    umask = None
    if rng.random() < 0.4:
        umask = (rng.random((H, W)) < 0.7).astype(np.uint8) * rng.integers(0, 256, (H, W)).astype(np.uint8)
    zs = float(rng.choice([5000.0, 1000.0, 1.0]))
    aX, _ = pp.get_aX(bgr, depth, *K, z_scaling=zs, threshold=thr, mask_u8=umask)
    P.set_ref_frame(bgr, depth, z_scaling=zs, threshold=thr, mask=umask)
    want = aX[:3].T
    got = P.get_points() if P.num_points else np.zeros((0, 3))
    same = P.num_points == aX.shape[1] and (np.array_equal(got, want) or np.array_equal(got, want.astype(np.float32).astype(np.float64)))
    check("lap.points", same, info + (thr, zs, umask is not None))
    # -- Canny flavour
    lo, hi = float(rng.uniform(0, 200)), float(rng.uniform(0, 300))
    edges = pp.canny_u8(pp.rgb2gray_u8(pp.box_blur3_u8(bgr)), lo, hi)
    norm = [(0.0, 1.0), (0.0, 255.0), None][int(rng.integers(3))]
    cmask = umask if rng.random() < 0.5 else None
    gotc = P.set_now_frame_canny(bgr, mask=cmask, low=lo, high=hi, normalize=norm, debug=True)
    keep = (edges != 0) if cmask is None else ((edges != 0) & (cmask > 1))     # the _masked variants keep edges where mask > 1
    check("canny.edges", np.array_equal(gotc["edges"] != 0, keep) and set(np.unique(gotc["edges"])) <= {0, 255}, info + (lo, hi, cmask is not None))
    if keep.any():
        check("canny.chamfer", np.array_equal(gotc["chamfer"], pp.chamfer3x3_fixed(keep)), info + (lo, hi, cmask is not None))
    gotd = P.set_now_frame_canny(bgr, mask=cmask, normalize=norm, debug=True)
    check("canny.dt", np.array_equal(gotd["dt"], pp.get_distance_transform2(bgr, mask_u8=cmask, normalize=norm)), info + (norm, cmask is not None))
    aXc, _ = pp.get_aX_canny(bgr, depth, *K, z_scaling=zs)
    P.set_ref_frame_canny(bgr, depth, z_scaling=zs)
    want = aXc[:3].T
    got = P.get_points() if P.num_points else np.zeros((0, 3))
    check("canny.points", P.num_points == aXc.shape[1] and (np.array_equal(got, want) or np.array_equal(got, want.astype(np.float32).astype(np.float64))), info)
    # -- ROS flavour (L2 gradient Canny on the colour image, exact EDT), optionally after the node's half-resolution steps
    t1, t2 = float(rng.uniform(20, 400)), float(rng.uniform(20, 400))
    hv = int(rng.integers(0, 3)) if min(H, W) >= 24 else 0
    Hr, Wr = H - H % (1 << hv), W - W % (1 << hv)     # the device resize wants extents divisible by 2^halvings
    bgr = np.ascontiguousarray(bgr[:Hr, :Wr]); H, W = Hr, Wr
    small, df = bgr, (rng.random((H, W)) * 4.0 + 0.4).astype(np.float32)
    df[rng.random((H, W)) < 0.2] = 0.0
    if hv and rng.random() < 0.5:
        df[rng.random((H, W)) < 0.05] = np.nan      # the node zeroes NaN depth before its resize (src/ea.cpp:56-62)
    dsmall = df
    for _ in range(hv):
        small, dsmall = pp.resize_half_bgr8(small), pp.resize_half_f32(dsmall)
    e2 = pp.canny_u8(small, t1, t2, l2_gradient=True)
    if (e2 > 0).any():
        want_dt = pp.ros_now_distance_transform(small, t1, t2)
        if hv == 0:
            gotr = P.set_now_frame_ros(bgr, t1=t1, t2=t2, debug=True)
            check("ros.edges", np.array_equal(gotr["edges"], e2), info + (t1, t2, hv))
            check("ros.dt", np.array_equal(gotr["dt"], want_dt), info + (t1, t2, hv))
        else:
            P.set_now_frame_ros(bgr, t1=t1, t2=t2, halvings=hv)
            check("ros.dt.halved", np.array_equal(P.get_dt().astype(np.float32), want_dt), info + (t1, t2, hv))
        pts, _ = pp.ros_ref_points(small, dsmall, *K, low=t1, high=t2)
        P.set_ref_frame_ros(bgr, df, t1=t1, t2=t2, halvings=hv)
        got = P.get_points() if P.num_points else np.zeros((0, 3))
        check("ros.points", P.num_points == pts.shape[1] and (np.array_equal(got, pts.T) or np.array_equal(got, pts.T.astype(np.float32).astype(np.float64))), info + (t1, t2, hv))
    else:
        try:
            P.set_now_frame_ros(bgr, t1=t1, t2=t2, halvings=hv)
            check("ros.noedge_refused", False, info)
        except capi.EAError:
            check("ros.noedge_refused", True, info)
    P.close()
    n_cases += 1
print("soak (pre-processing) %s: %d frames, seed %d; checks %s" % ("ok" if not n_fail else "FAILED %d" % n_fail, n_cases, seed, counts))

"""SURVEY §8(f) rows 1-2 on the GPU: edge-point extractor (get_aX, standalone/utils.cpp:201-281) and DT
producer (get_distance_transform, utils.cpp:38-83).  Integer / byte / index work: BIT-EXACT against the
numpy restatement (oracle/preprocess_np.py); the one reference-held number, 44457 edge points = 1482
residual blocks at stride 30 (standalone/README.md:34), is reproduced by the GPU path too."""
import os

import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "rgbd")
K = (525.0, 525.0, 319.5, 239.5)


@pytest.fixture(scope="module")
def frames():
    from oracle import preprocess_np as pp
    return dict(pp=pp,
                rgb1=pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png")), depth1=pp.load_depth_u16(os.path.join(G, "depth_1.png")),
                rgb3=pp.load_rgb_as_bgr(os.path.join(G, "rgb_3.png")), rgb5=pp.load_rgb_as_bgr(os.path.join(G, "rgb_5.png")))


@pytest.mark.parametrize("name", ["rgb3", "rgb5", "rgb1"])
@pytest.mark.parametrize("median,normalize", [(True, True), (False, False)])
def test_dt_producer_stages_bit_exact(hip, frames, name, median, normalize):
    pp = frames["pp"]
    img = frames[name]
    P = hip.Problem(*K, dtype=hip.EA_F64)
    st = P.set_now_frame(img, threshold=35, median=median, normalize=normalize, debug=True)
    lap = pp.edge_strength(img)
    assert np.array_equal(st["lap"], lap)
    B = np.where(lap > 35, 0, 255).astype(np.uint8)
    mask = pp.median_blur3_u8(B) if median else B
    assert np.array_equal(st["mask"], mask)
    cham = pp.chamfer3x3_fixed(mask == 0)  # OpenCV's two-pass raster chamfer
    assert np.array_equal(st["chamfer"].astype(np.int64), cham)  # == closed-form shortest-path search on the GPU
    dist = pp.distance_transform_l2_3(mask)
    want = pp.normalize_minmax_f32(dist) if normalize else dist
    assert st["dt"].dtype == np.float32 and np.array_equal(st["dt"], want)
    # and what the problem holds in HBM is that image (fp32 values are exact in the fp64 problem)
    assert np.array_equal(P.get_dt(), want.astype(np.float64))
    P.close()


def test_dt_producer_edge_cases(hip, frames):
    pp = frames["pp"]
    rng = np.random.default_rng(3)
    # tiny, odd-sized, and feature-free images
    for (H, W) in [(3, 3), (5, 17), (37, 64), (64, 257)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        P = hip.Problem(*K, dtype=hip.EA_F32)
        st = P.set_now_frame(img, threshold=35, median=True, normalize=True, debug=True)
        lap = pp.edge_strength(img)
        mask = pp.median_blur3_u8(np.where(lap > 35, 0, 255).astype(np.uint8))
        assert np.array_equal(st["lap"], lap) and np.array_equal(st["mask"], mask)
        assert np.array_equal(st["chamfer"].astype(np.int64), pp.chamfer3x3_fixed(mask == 0))
        P.close()
    flat = np.full((40, 50, 3), 77, dtype=np.uint8)  # no edges at all: every distance saturates
    P = hip.Problem(*K, dtype=hip.EA_F32)
    st = P.set_now_frame(flat, debug=True, normalize=False)
    assert (st["mask"] == 255).all() and (st["chamfer"] == 2 ** 30 - 1).all()
    P.close()


@pytest.mark.parametrize("dtype_name", ["f64", "f32"])
def test_edge_point_extractor_bit_exact(hip, frames, dtype_name):
    pp = frames["pp"]
    aX, (vv, uu) = pp.get_aX(frames["rgb1"], frames["depth1"], *K)
    P = hip.Problem(*K, dtype=hip.EA_F64 if dtype_name == "f64" else hip.EA_F32)
    P.set_ref_frame(frames["rgb1"], frames["depth1"], z_scaling=5000.0, threshold=35)
    assert P.num_points == 44457 == aX.shape[1]          # the reference log's 1482 blocks x stride 30
    assert -(-P.num_points // 30) == 1482
    xyz = P.get_points()
    want = aX[:3].T
    if dtype_name == "f64":
        assert np.array_equal(xyz, want)                  # same doubles, same raster order
    else:
        assert np.array_equal(xyz, want.astype(np.float32).astype(np.float64))
    P.close()


def test_edge_point_extractor_edge_cases(hip, frames):
    pp = frames["pp"]
    rng = np.random.default_rng(4)
    for (H, W) in [(3, 3), (31, 33), (100, 1030)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        depth = rng.integers(0, 3, (H, W)).astype(np.uint16) * rng.integers(0, 20000, (H, W)).astype(np.uint16)
        aX, _ = pp.get_aX(img, depth, *K, z_scaling=5000.0, threshold=35)
        P = hip.Problem(*K, dtype=hip.EA_F64)
        P.set_ref_frame(img, depth)
        assert P.num_points == aX.shape[1]
        assert np.array_equal(P.get_points(), aX[:3].T)
        P.close()
    P = hip.Problem(*K, dtype=hip.EA_F64)   # no depth anywhere -> empty cloud
    P.set_ref_frame(frames["rgb1"], np.zeros_like(frames["depth1"]))
    assert P.num_points == 0
    P.close()


@pytest.mark.parametrize("b", [3, 5])
def test_frames_to_pose_end_to_end(hip, frames, golden, b):
    """raw frames -> GPU edge points + GPU DT -> device LM: the whole of edge_align_test1's compute
    (standalone_edge_align.cpp:170-286, stride 1) without the CPU touching a pixel"""
    P = hip.Problem(*K, dtype=hip.EA_F64)
    P.set_ref_frame(frames["rgb1"], frames["depth1"])
    P.set_now_frame(frames["rgb%d" % b])
    q, t, s = P.solve([1, 0, 0, 0], [0, 0, 0])
    tag = "b%d_s1" % b
    assert s["num_iterations"] == int(golden[tag + "_lm_iterations"]) and s["why"] == str(golden[tag + "_lm_why"])
    assert synth.rotation_angle_between(q, golden[tag + "_lm_q"]) < 1e-7
    assert np.linalg.norm(t - golden[tag + "_lm_t"]) < 1e-7
    P.close()


def test_c4_batch_of_frame_pairs(hip, oracle):
    """C4 (BASELINE.json configs[3]): the TUM fr1_desk sequence is not available, so the batch is
    synthesised as prescribed — the 20 ordered pairs of the 5 bundled frames x seeded initial-pose
    perturbations (rotation <= 1 deg, translation <= 2 cm, seed 4) — 40 problems here (32 per GPU in the
    8-GPU layout).  Everything from raw frames to poses runs on the device; the batch must follow the
    single-problem solves, and sampled problems must land on the oracle's pose."""
    from oracle import preprocess_np as pp
    rgb = {k: pp.load_rgb_as_bgr(os.path.join(G, "rgb_%d.png" % k)) for k in range(1, 6)}
    dep = {k: pp.load_depth_u16(os.path.join(G, "depth_%d.png" % k)) for k in range(1, 6)}
    rng = np.random.default_rng(4)
    pairs = [(a, b) for a in range(1, 6) for b in range(1, 6) if a != b]
    Ps, q0s, t0s, meta = [], [], [], []
    for (a, b) in pairs:
        for rep in range(2):
            P = hip.Problem(*K, dtype=hip.EA_F64)
            P.set_ref_frame(rgb[a], dep[a])
            P.set_now_frame(rgb[b])
            ax = rng.normal(size=3)
            q0 = synth.quat_from_axis_angle(ax, np.deg2rad(rng.uniform(0, 1.0))) if rep else np.array([1.0, 0, 0, 0])
            t0 = rng.uniform(-0.02, 0.02, 3) / np.sqrt(3) if rep else np.zeros(3)
            Ps.append(P); q0s.append(q0); t0s.append(t0); meta.append((a, b))
    B = hip.Batch(Ps)
    q, t, ss = B.solve(np.array(q0s), np.array(t0s))
    assert all(s["termination"] in (hip.CONVERGENCE, hip.NO_CONVERGENCE) for s in ss)
    for i in (0, 7, 13, 22, 39):   # batch == single up to the summation order (the launch shape differs: 2 points per lane)
        q1, t1, s1 = Ps[i].solve(q0s[i], t0s[i])
        assert np.abs(q1 - q[i]).max() < 1e-9 and np.abs(t1 - t[i]).max() < 1e-9 and s1["num_iterations"] == ss[i]["num_iterations"]
    for i in (1, 18, 31):          # and the oracle (numpy pre-processing + C LM) lands on the same pose
        a, b = meta[i]
        aX, _ = pp.get_aX(rgb[a], dep[a], *K)
        O = oracle.OracleProblem(pp.grid_view_of_image(pp.get_distance_transform(rgb[b])), *K)
        qo, to, so = O.solve(aX[:3].T.copy(), q0s[i], t0s[i])
        assert synth.rotation_angle_between(q[i], qo) < 1e-4 and np.linalg.norm(t[i] - to) < 1e-3
        assert ss[i]["num_iterations"] == so["num_iterations"]
    B.close()
    for P in Ps:
        P.close()


# ---- Canny flavour (get_distance_transform2* / get_aX_canny, ref: utils.cpp:85-199, :371-462) ----------------

def _random_frame(seed, H=97, W=131):
    """smooth blobs + sharp steps + noise: plenty of weak/strong Canny candidates and long thin chains"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((H, W, 3))
    for c in range(3):
        a = 60 * np.sin(xx / (5 + 3 * c) + rng.random() * 6) + 60 * np.cos(yy / (7 + c) + rng.random() * 6)
        a += 80 * ((xx - W * rng.random()) ** 2 + (yy - H * rng.random()) ** 2 < (15 + 10 * c) ** 2)
        a += 50 * (xx > W * rng.random()) + rng.normal(0, 6, (H, W))
        img[:, :, c] = a
    img -= img.min()
    return (img * (255.0 / img.max())).astype(np.uint8)


def test_canny_edges_and_dt2_bit_exact(hip, frames):
    """Every stage of the Canny flavour against the OpenCV-semantics restatement: edge map (blur 3x3, gray, Sobel,
    non-maximum suppression, hysteresis) and chamfer distance as integers, the float DT bit for bit."""
    pp = frames["pp"]
    for k, bgr in enumerate([frames["rgb3"], frames["rgb1"], _random_frame(1), _random_frame(2, 40, 500), _random_frame(3, 333, 35)]):
        P = hip.Problem(525.0, 525.0, 319.5, 239.5, dtype=hip.EA_F64)
        got = P.set_now_frame_canny(bgr, debug=True)
        edges = pp.canny_edges_of_frame(bgr)
        assert np.array_equal(got["edges"], edges), k
        assert np.array_equal(got["chamfer"], pp.chamfer3x3_fixed(edges != 0)), k
        want = pp.get_distance_transform2(bgr)
        assert np.array_equal(got["dt"], want), k
        assert np.array_equal(P.get_dt().astype(np.float32), want), k
        # other thresholds (swapped on purpose: cv::Canny orders them itself)
        got = P.set_now_frame_canny(bgr, low=120.0, high=45.5, debug=True)
        assert np.array_equal(got["edges"], pp.canny_u8(pp.rgb2gray_u8(pp.box_blur3_u8(bgr)), 120.0, 45.5)), k
        P.close()


def test_canny_dt2_variants(hip, frames):
    """_NoNormalize, _masked (normalised to [0, 255]) and _masked_NoNormalize (ref: utils.cpp:108-199)."""
    pp = frames["pp"]
    bgr = frames["rgb3"]
    H, W = bgr.shape[:2]
    rng = np.random.default_rng(9)
    mask = np.zeros((H, W), np.uint8)
    mask[60:400, 100:560] = 255
    mask[rng.random((H, W)) < 0.05] = 1     # 1 is NOT kept: the reference thresholds at > 1
    mask[200:230, 300:340] = 0
    P = hip.Problem(525.0, 525.0, 319.5, 239.5, dtype=hip.EA_F32)
    got = P.set_now_frame_canny(bgr, normalize=None, debug=True)
    assert np.array_equal(got["dt"], pp.get_distance_transform2(bgr, normalize=None))
    got = P.set_now_frame_canny(bgr, mask=mask, normalize=(0.0, 255.0), debug=True)
    assert np.array_equal(got["dt"], pp.get_distance_transform2(bgr, mask_u8=mask, normalize=(0.0, 255.0)))
    assert np.array_equal(P.get_dt().astype(np.float32), got["dt"])
    got = P.set_now_frame_canny(bgr, mask=mask, normalize=None, debug=True)
    assert np.array_equal(got["dt"], pp.get_distance_transform2(bgr, mask_u8=mask, normalize=None))
    # an image without a single edge: every distance saturates alike, the normalised DT is constant
    flat = np.full((64, 80, 3), 117, np.uint8)
    got = P.set_now_frame_canny(flat, debug=True)
    assert got["edges"].max() == 0 and np.array_equal(got["dt"], pp.get_distance_transform2(flat))
    P.close()


def test_canny_edge_points_bit_exact(hip, frames):
    """get_aX_canny: Canny edge && depth > 0, raster order, fp64 back-projection."""
    pp = frames["pp"]
    for bgr, depth in ((frames["rgb1"], frames["depth1"]),):
        aX, _ = pp.get_aX_canny(bgr, depth, *K)
        P = hip.Problem(*K, dtype=hip.EA_F64)
        P.set_ref_frame_canny(bgr, depth)
        assert P.num_points == aX.shape[1] > 10000
        assert np.array_equal(P.get_points(), aX[:3].T)
        P.close()


def test_canny_frames_to_pose_end_to_end(hip, oracle, frames):
    """The compute of the reference's later tests (standalone_edge_align.cpp:372-490: get_aX_canny-style points,
    get_distance_transform2, CauchyLoss, LM) from raw frames, against the oracle's solve on the restated inputs."""
    pp = frames["pp"]
    aX, _ = pp.get_aX_canny(frames["rgb1"], frames["depth1"], *K)
    dt = pp.get_distance_transform2(frames["rgb3"])
    O = oracle.OracleProblem(pp.grid_view_of_image(dt), *K, loss=oracle.LOSS_CAUCHY, loss_a=1.0)
    qo, to, so = O.solve(aX[:3].T.copy(), np.array([1.0, 0, 0, 0]), np.zeros(3))
    P = hip.Problem(*K, dtype=hip.EA_F64)
    P.set_loss(hip.LOSS_CAUCHY, 1.0)
    P.set_ref_frame_canny(frames["rgb1"], frames["depth1"])
    P.set_now_frame_canny(frames["rgb3"])
    q, t, s = P.solve(np.array([1.0, 0, 0, 0]), np.zeros(3))
    assert s["num_iterations"] == so["num_iterations"] and s["why"] == so["why"]
    assert np.abs(q - qo).max() < 1e-9 and np.abs(t - to).max() < 1e-9
    assert s["final_cost"] == pytest.approx(so["final_cost"], rel=1e-10)
    P.close()


def test_masked_edge_points_bit_exact(hip, frames):
    """get_aX_mask (ref: utils.cpp:283-369): gradient > 35 && depth > 0 && mask > 0."""
    pp = frames["pp"]
    H, W = frames["depth1"].shape
    rng = np.random.default_rng(11)
    mask = (rng.random((H, W)) < 0.6).astype(np.uint8) * rng.integers(1, 256, (H, W)).astype(np.uint8)
    mask[100:300, 150:500] = 0
    aX, _ = pp.get_aX(frames["rgb1"], frames["depth1"], *K, mask_u8=mask)
    P = hip.Problem(*K, dtype=hip.EA_F64)
    P.set_ref_frame(frames["rgb1"], frames["depth1"], mask=mask)
    assert 0 < P.num_points == aX.shape[1] < 44457
    assert np.array_equal(P.get_points(), aX[:3].T)
    P.set_ref_frame(frames["rgb1"], frames["depth1"], mask=np.zeros((H, W), np.uint8))
    assert P.num_points == 0
    P.close()


# ---- ROS flavour (SolveEA::setRefFrame / setNowFrame, ref: src/SolveEA.cpp:29-119) ---------------------------

def test_ros_producers_bit_exact(hip, frames):
    """Canny(bgr, 150, 100, 3, true) on the 3-channel image, exact Euclidean DT normalised to [0, 255], and the
    float-depth back-projection with Z == 0 -> 1 -- every stage against the restatement."""
    pp = frames["pp"]
    Kh = (0.5 * 525.0, 0.5 * 525.0, 0.5 * 319.5, 0.5 * 239.5)  # src/SolveEA.cpp:15-18: half-resolution intrinsics
    for k, bgr in enumerate([frames["rgb3"], frames["rgb1"][::2, ::2].copy(), _random_frame(5, 120, 160)]):
        P = hip.Problem(*Kh, dtype=hip.EA_F64)
        got = P.set_now_frame_ros(bgr, debug=True)
        edges = pp.canny_u8(bgr, 150.0, 100.0, l2_gradient=True)
        assert np.array_equal(got["edges"], edges), k
        want = pp.ros_now_distance_transform(bgr)
        assert np.array_equal(got["dt"], want), k
        assert got["dt"].min() == 0.0 and got["dt"].max() == 255.0
        assert np.array_equal(P.get_dt().astype(np.float32), want), k
        H, W = bgr.shape[:2]
        rng = np.random.default_rng(20 + k)
        depth = (rng.random((H, W)) * 4.0 + 0.4).astype(np.float32)
        depth[rng.random((H, W)) < 0.2] = 0.0        # invalid depth: the reference substitutes Z = 1
        pts, _ = pp.ros_ref_points(bgr, depth, *Kh)
        P.set_ref_frame_ros(bgr, depth)
        assert P.num_points == pts.shape[1] == int((edges > 0).sum())
        assert np.array_equal(P.get_points(), pts.T)
        P.close()
    P = hip.Problem(*Kh, dtype=hip.EA_F32)
    with pytest.raises(Exception):
        P.set_now_frame_ros(np.full((48, 64, 3), 90, np.uint8))   # no edge: undefined upstream, refused here
    P.close()


@pytest.mark.parametrize("flavour", [0, 1])
def test_frame_to_frame_tracker(hip, flavour):
    """ea_tracker: the bundled five frames as a sequence.  Each push must equal the manual set_now / solve-from-prior /
    set_ref sequence on a plain problem (same calls underneath), and the relative motions between consecutive grabs
    of the hand-held sequence stay small."""
    from oracle import preprocess_np as pp
    seq = [(pp.load_rgb_as_bgr(os.path.join(G, "rgb_%d.png" % i)), pp.load_depth_u16(os.path.join(G, "depth_%d.png" % i))) for i in range(1, 6)]
    T = hip.Tracker(*K, dtype=hip.EA_F64, flavour=flavour, loss=(hip.LOSS_CAUCHY, 1.0))
    P = hip.Problem(*K, dtype=hip.EA_F64)
    P.set_loss(hip.LOSS_CAUCHY, 1.0)
    q_prior, t_prior = np.array([1.0, 0, 0, 0]), np.zeros(3)
    for k, (bgr, depth) in enumerate(seq):
        q, t, s = T.push_frame(bgr, depth)
        if k == 0:
            assert s is None and np.array_equal(q, [1, 0, 0, 0]) and np.array_equal(t, [0, 0, 0])
        else:
            (P.set_now_frame if flavour == 0 else P.set_now_frame_canny)(bgr)
            qm, tm, sm = P.solve(q_prior, t_prior)
            assert np.array_equal(q, qm) and np.array_equal(t, tm) and s["num_iterations"] == sm["num_iterations"]
            assert s["termination"] != 2 and s["final_cost"] < s["initial_cost"]
            assert 2 * np.arccos(min(1.0, abs(q[0]))) < 0.2 and np.linalg.norm(t) < 0.3
            q_prior, t_prior = qm, tm
        (P.set_ref_frame if flavour == 0 else P.set_ref_frame_canny)(bgr, depth)
    T.close(); P.close()


def test_frames_in_page_locked_memory(hip):
    """ea_host_alloc: frames handed over from page-locked memory (direct DMA instead of the runtime's staged copy) give
    what the same frames give from ordinary memory, bit for bit; the block is plain host memory and can be freed."""
    from oracle import preprocess_np as pp
    seq = [(pp.load_rgb_as_bgr(os.path.join(G, "rgb_%d.png" % i)), pp.load_depth_u16(os.path.join(G, "depth_%d.png" % i))) for i in range(1, 4)]
    TA, TB = hip.Tracker(*K, dtype=hip.EA_F64, loss=(hip.LOSS_CAUCHY, 1.0)), hip.Tracker(*K, dtype=hip.EA_F64, loss=(hip.LOSS_CAUCHY, 1.0))
    for bgr, depth in seq:
        pb = hip.pinned_array(bgr.shape, bgr.dtype); pb[...] = bgr
        pd = hip.pinned_array(depth.shape, depth.dtype); pd[...] = depth
        qa, ta, sa = TA.push_frame(bgr, depth)
        qb, tb, sb = TB.push_frame(pb, pd)
        assert np.array_equal(qa, qb) and np.array_equal(ta, tb)
        assert (sa is None) == (sb is None) and (sa is None or sa["final_cost"] == sb["final_cost"])
        del pb, pd   # (freed: the library copied what it needs)
    TA.close(); TB.close()
    L = hip.load()
    import ctypes as C
    L.ea_host_alloc.restype = C.c_void_p
    L.ea_host_alloc.argtypes = [C.c_size_t, C.c_int]
    assert L.ea_host_alloc(0, 0) is None and b"zero" in L.ea_last_error()
    assert L.ea_host_alloc(64, 9999) is None


def test_random_frame_sizes_all_flavours(hip, frames):
    """Widths / heights on both sides of the kernels' tile sizes (16, 32, 64, 256), tiny frames, thin strips, dense
    noise: every producer of the three flavours bit for bit against the restatement; frames with an extent below 3
    are refused (no 3x3 neighbourhood to speak of)."""
    pp = frames["pp"]
    rng = np.random.default_rng(99)
    sizes = [(3, 3), (4, 9), (7, 33), (31, 31), (32, 32), (33, 65), (65, 63), (3, 300), (300, 3), (129, 255), (130, 257), (17, 513)]
    sizes += [(int(rng.integers(3, 200)), int(rng.integers(3, 300))) for _ in range(6)]
    for k, (H, W) in enumerate(sizes):
        bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if k % 3 == 0 else _random_frame(100 + k, H, W)
        depth = rng.integers(0, 30000, (H, W)).astype(np.uint16)
        depth[rng.random((H, W)) < 0.15] = 0
        P = hip.Problem(*K, dtype=hip.EA_F64)
        st = P.set_now_frame(bgr, threshold=35, median=True, normalize=True, debug=True)
        lap = pp.edge_strength(bgr)
        mask = pp.median_blur3_u8(np.where(lap > 35, 0, 255).astype(np.uint8))
        assert np.array_equal(st["lap"], lap) and np.array_equal(st["mask"], mask), (H, W)
        if (mask == 0).any():
            assert np.array_equal(st["chamfer"].astype(np.int64), pp.chamfer3x3_fixed(mask == 0)), (H, W)
            assert np.array_equal(st["dt"], pp.get_distance_transform(bgr)), (H, W)
        aX, _ = pp.get_aX(bgr, depth, *K)
        P.set_ref_frame(bgr, depth)
        assert P.num_points == aX.shape[1] and (P.num_points == 0 or np.array_equal(P.get_points(), aX[:3].T)), (H, W)
        got = P.set_now_frame_canny(bgr, debug=True)
        edges = pp.canny_edges_of_frame(bgr)
        assert np.array_equal(got["edges"], edges), (H, W)
        if (edges != 0).any():
            assert np.array_equal(got["chamfer"], pp.chamfer3x3_fixed(edges != 0)), (H, W)
            assert np.array_equal(got["dt"], pp.get_distance_transform2(bgr)), (H, W)
        aXc, _ = pp.get_aX_canny(bgr, depth, *K)
        P.set_ref_frame_canny(bgr, depth)
        assert P.num_points == aXc.shape[1] and (P.num_points == 0 or np.array_equal(P.get_points(), aXc[:3].T)), (H, W)
        e2 = pp.canny_u8(bgr, 150.0, 100.0, l2_gradient=True)
        if (e2 > 0).any():
            got = P.set_now_frame_ros(bgr, debug=True)
            assert np.array_equal(got["edges"], e2) and np.array_equal(got["dt"], pp.ros_now_distance_transform(bgr)), (H, W)
            df = (rng.random((H, W)) * 4.0 + 0.4).astype(np.float32)
            df[rng.random((H, W)) < 0.2] = 0.0
            pts, _ = pp.ros_ref_points(bgr, df, *K)
            P.set_ref_frame_ros(bgr, df)
            assert P.num_points == pts.shape[1] and np.array_equal(P.get_points(), pts.T), (H, W)
        P.close()
    P = hip.Problem(*K, dtype=hip.EA_F64)
    for (H, W) in [(1, 40), (40, 2)]:
        with pytest.raises(Exception):
            P.set_now_frame(np.zeros((H, W, 3), np.uint8))
    P.close()


# ---- the node's half-resolution step on the device (src/ea.cpp:38, :56-62) ------------------------------------------

def test_resize_half_bit_exact(hip):
    from oracle import preprocess_np as pp
    rng = np.random.default_rng(21)
    G = os.path.join(ROOT, "tests", "golden", "rgbd")
    frames = [pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png")), rng.integers(0, 256, (62, 94, 3), dtype=np.uint8),
              np.full((4, 6, 3), 255, np.uint8)]
    for im in frames:
        assert np.array_equal(hip.resize_half(im), pp.resize_half_bgr8(im))
    d = pp.load_depth_u16(os.path.join(G, "depth_1.png")).astype(np.float32) / np.float32(5000.0)
    d2 = d.copy(); d2[rng.random(d.shape) < 0.05] = np.nan           # the callback's invalid depths
    for dep, nz in ((d, True), (d2, True), (rng.normal(size=(30, 44)).astype(np.float32), False)):
        got, want = hip.resize_half(dep, nan_to_zero=nz), pp.resize_half_f32(dep, nan_to_zero=nz)
        assert not np.isnan(got).any() and np.array_equal(got, want)
    with pytest.raises(hip.EAError):
        hip.resize_half(np.zeros((5, 6, 3), np.uint8))               # odd extent


def test_ros_producers_from_full_resolution_frames(hip, oracle):
    """Full-resolution frames -> x0.5 on the device -> ROS producers -> DOGLEG solve, against the same chain with the
    x0.5 done by the numpy restatement on the host: identical points, identical DT, identical pose.  Two halvings build
    the next pyramid level the same way."""
    from oracle import preprocess_np as pp
    G = os.path.join(ROOT, "tests", "golden", "rgbd")
    ref = pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png"))
    now = pp.load_rgb_as_bgr(os.path.join(G, "rgb_2.png"))
    depth = pp.load_depth_u16(os.path.join(G, "depth_1.png")).astype(np.float32) / np.float32(5000.0)
    depth[::7, ::5] = np.nan
    for halvings in (1, 2):
        r, n, d = ref, now, depth
        for k in range(halvings):
            r, n = pp.resize_half_bgr8(r), pp.resize_half_bgr8(n)
            d = pp.resize_half_f32(d, nan_to_zero=(k == 0))
        s = 0.5 ** halvings
        Kl = (525.0 * s, 525.0 * s, 319.5 * s, 239.5 * s)          # src/SolveEA.cpp:15-18 scales all four
        A, B = hip.Problem(*Kl), hip.Problem(*Kl)
        for P in (A, B):
            P.set_flavour(z_guard=0.0, z_eps=0.001, rot_transposed=True); P.set_loss(hip.LOSS_TRIVIAL, 1.0)
        A.set_ref_frame_ros(ref, depth, halvings=halvings); A.set_now_frame_ros(now, halvings=halvings)
        B.set_ref_frame_ros(r, d); B.set_now_frame_ros(n)
        assert A.num_points == B.num_points > 1000
        assert np.array_equal(A.get_points(), B.get_points()) and np.array_equal(A.get_dt(), B.get_dt())
        qa, ta, sa = A.solve([1, 0, 0, 0], [0, 0, 0], strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
        qb, tb, sb = B.solve([1, 0, 0, 0], [0, 0, 0], strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
        assert np.array_equal(qa, qb) and np.array_equal(ta, tb) and sa["num_iterations"] == sb["num_iterations"]
        A.close(); B.close()
    with pytest.raises(hip.EAError):
        P = hip.Problem(*Kl)
        P.set_now_frame_ros(now[:478], halvings=2)                   # 478 is not divisible by 4

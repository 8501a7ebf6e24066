"""The C++ drop-in surface on the GPU: the reference's own call sequence
(standalone/standalone_edge_align.cpp:256-293 and src/ea.cpp:184-191) compiled against
edge_alignment_amd/include/{EAResidue.h, ceres/*, ros/SolveEA.h} and linked to libea_hip.so."""
import os
import struct
import subprocess

import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binaries(hip):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    return os.path.join(ROOT, "examples")


def _write_problem(path, aX, grid, K):
    # grid is the Grid2D view (W x H row-major) == the column-major H x W Eigen matrix's raw data
    W, H = grid.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", aX.shape[1], H, W))
        f.write(struct.pack("<dddd", *K))
        f.write(np.ascontiguousarray(aX.T, dtype=np.float64).tobytes())   # 4xN column-major
        f.write(np.ascontiguousarray(grid, dtype=np.float64).tobytes())


def test_edge_align_test1_call_sequence(binaries, bundled_pair, golden, tmp_path):
    p = str(tmp_path / "pair13.bin")
    _write_problem(p, bundled_pair["aX"], bundled_pair["grids"][3], bundled_pair["K"])
    for stride in (30, 1):
        out = subprocess.run([os.path.join(binaries, "standalone_test1"), p, str(stride)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        v = [float(x) for x in out.stdout.split()]
        q, t, iters, term = np.array(v[:4]), np.array(v[4:7]), int(v[7]), int(v[8])
        tag = "b3_s%d" % stride
        assert term == 0  # ceres::CONVERGENCE
        assert synth.rotation_angle_between(q, golden[tag + "_lm_q"]) < 1e-7
        assert np.linalg.norm(t - golden[tag + "_lm_t"]) < 1e-7
        # Ceres' "Minimizer iterations" = successful + unsuccessful steps; the last iteration, which
        # trips the function tolerance, is neither
        assert iters == int(golden[tag + "_lm_successful"])
        assert v[9] == pytest.approx(float(golden[tag + "_lm_it_cost"][0]), rel=1e-10)
        assert "Residual blocks" in out.stderr and "CONVERGENCE" in out.stderr
        # problem.Evaluate (src/SolveEA.cpp:241) at the initial pose: the cost the solve starts from, one
        # loss-corrected residual per block (1/2 sum r_c^2 is that cost for Cauchy only approximately: rho(s) <= s)
        assert int(v[11]) == 1 and v[12] == pytest.approx(v[9], rel=1e-12)
        assert int(v[13]) == -(-bundled_pair["aX"].shape[1] // stride) and int(v[15]) == 6
        assert 0.0 < v[14] <= v[12] * (1 + 1e-12)
        # Evaluate with the Jacobian handed out (ceres::CRSMatrix, dense 1x6 rows): one row per block, J^T r = gradient
        assert int(v[16]) == int(v[13]) and v[17] < 1e-10


def test_solve_ea_class_dogleg(binaries, oracle, bundled_pair, tmp_path):
    p = str(tmp_path / "pair13.bin")
    _write_problem(p, bundled_pair["aX"], bundled_pair["grids"][3], bundled_pair["K"])
    out = subprocess.run([os.path.join(binaries, "solve_ea_demo"), p], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    v = [float(x) for x in out.stdout.strip().split("\n")[-1].split()]
    q, t = np.array(v[:4]), np.array(v[4:7])
    # SolveEA() hard-codes half-resolution TUM intrinsics, no loss, R^T, z+0.001, DOGLEG, 25 iterations
    O = oracle.OracleProblem(bundled_pair["grids"][3], 262.5, 262.5, 159.75, 119.75, loss=oracle.LOSS_TRIVIAL,
                             z_guard=0.0, z_eps=0.001, rot_transposed=True)
    X = bundled_pair["aX"][:3, ::10].T.copy()
    qo, to, so = O.solve(X, [1, 0, 0, 0], [0, 0, 0], strategy=oracle.STRATEGY_DOGLEG, max_num_iterations=25)
    assert synth.rotation_angle_between(q, qo) < 1e-7 and np.linalg.norm(t - to) < 1e-7
    assert int(v[7]) == so["termination"]


def test_ros_frames_through_solve_ea(binaries, oracle, tmp_path):
    """src/ea.cpp:184-191 on raw frames: SolveEA::setRefFrame / setNowFrame (raw-buffer overloads, pre-processing on
    the GPU) + setAsCERESProblem, against the oracle's DOGLEG solve on the restated pre-processing."""
    from oracle import preprocess_np as pp
    G = os.path.join(ROOT, "tests", "golden", "rgbd")
    # the node works at half resolution (src/ea.cpp:38, :62); depth in metres as float32, invalid = 0
    ref = pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png"))[::2, ::2].copy()
    now = pp.load_rgb_as_bgr(os.path.join(G, "rgb_2.png"))[::2, ::2].copy()
    depth = (pp.load_depth_u16(os.path.join(G, "depth_1.png"))[::2, ::2].astype(np.float32) / np.float32(5000.0)).copy()
    rows, cols = depth.shape
    path = str(tmp_path / "frames.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", rows, cols))
        f.write(ref.tobytes()); f.write(depth.tobytes()); f.write(now.tobytes())
    out = subprocess.run([os.path.join(binaries, "ros_frames_demo"), path], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    v = [float(x) for x in out.stdout.strip().split("\n")[-1].split()]
    q, t = np.array(v[:4]), np.array(v[4:7])
    Kh = (262.5, 262.5, 159.75, 119.75)
    pts, _ = pp.ros_ref_points(ref, depth, *Kh)
    assert int(v[8]) == pts.shape[1] > 5000
    dt = pp.ros_now_distance_transform(now)
    O = oracle.OracleProblem(pp.grid_view_of_image(dt), *Kh, loss=oracle.LOSS_TRIVIAL, z_guard=0.0, z_eps=0.001,
                             rot_transposed=True)
    qo, to, so = O.solve(pts.T.copy(), [1, 0, 0, 0], [0, 0, 0], strategy=oracle.STRATEGY_DOGLEG, max_num_iterations=25)
    assert synth.rotation_angle_between(q, qo) < 1e-7 and np.linalg.norm(t - to) < 1e-7
    assert int(v[7]) == so["termination"]


def test_ros_frames_full_resolution_resized_on_device(binaries, oracle, tmp_path):
    """The same call sequence fed with the frames as the callbacks receive them (640 x 480): SolveEA::setHalvings(1)
    moves the node's cv::resize(..., 0.5, 0.5) (src/ea.cpp:38, :56-62) onto the device; frames, edge points and distance
    transform never come back to the host between setRefFrame and the pose.  Checked against the oracle's DOGLEG solve on
    the restated pre-processing (numpy resize, Canny, exact EDT)."""
    from oracle import preprocess_np as pp
    G = os.path.join(ROOT, "tests", "golden", "rgbd")
    ref = pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png"))
    now = pp.load_rgb_as_bgr(os.path.join(G, "rgb_2.png"))
    depth = (pp.load_depth_u16(os.path.join(G, "depth_1.png")).astype(np.float32) / np.float32(5000.0)).copy()
    rows, cols = depth.shape
    path = str(tmp_path / "frames_full.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", rows, cols))
        f.write(ref.tobytes()); f.write(depth.tobytes()); f.write(now.tobytes())
    out = subprocess.run([os.path.join(binaries, "ros_frames_demo"), path, "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    v = [float(x) for x in out.stdout.strip().split("\n")[-1].split()]
    q, t = np.array(v[:4]), np.array(v[4:7])
    Kh = (262.5, 262.5, 159.75, 119.75)
    ref_h, now_h, depth_h = pp.resize_half_bgr8(ref), pp.resize_half_bgr8(now), pp.resize_half_f32(depth)
    pts, _ = pp.ros_ref_points(ref_h, depth_h, *Kh)
    assert int(v[8]) == pts.shape[1] > 5000
    dt = pp.ros_now_distance_transform(now_h)
    O = oracle.OracleProblem(pp.grid_view_of_image(dt), *Kh, loss=oracle.LOSS_TRIVIAL, z_guard=0.0, z_eps=0.001,
                             rot_transposed=True)
    qo, to, so = O.solve(pts.T.copy(), [1, 0, 0, 0], [0, 0, 0], strategy=oracle.STRATEGY_DOGLEG, max_num_iterations=25)
    assert synth.rotation_angle_between(q, qo) < 1e-7 and np.linalg.norm(t - to) < 1e-7
    assert int(v[7]) == so["termination"]


@pytest.mark.parametrize("ex", [False, True, "test7"])
def test_stereo_call_sequence(binaries, oracle, tmp_path, ex):
    """standalone_edge_align.cpp:778-815 (EAResidue + EAResidueSecondCam, CauchyLoss), :3195-3233
    (EAResidueEx + EAResidueSecondCamEx, TrivialLoss, 100 iterations) and :2590-2626 (tests 7-8: the plain functors,
    TrivialLoss, 100 iterations, every iterStep-th point with iterStep = N / 1000) through the drop-in headers"""
    test7, ex = ex == "test7", ex is True
    K1 = (130.0, 132.0, 79.5, 59.5)
    K2 = (128.0, 129.0, 81.0, 58.0)
    dist = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    T12 = synth.rigid_4x4(synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.04), [0.11, 0.004, -0.012])
    Q = synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0))
    T = np.array([0.01, -0.005, 0.02])
    fams = synth.make_stereo_problem(120, 160, 3000, 2000, 9, K1, K2, T12, Q, T, distortion=dist if ex else None)
    p = str(tmp_path / "stereo.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<iiii", 3000, 2000, 120, 160))
        f.write(struct.pack("<dddd", *K1)); f.write(struct.pack("<dddd", *K2)); f.write(struct.pack("<ddddd", *dist))
        f.write(np.ascontiguousarray(T12, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(np.linalg.inv(T12), dtype=np.float64).tobytes())
        for fam in fams:
            aX = np.concatenate([fam["xyz"], np.ones((fam["xyz"].shape[0], 1))], axis=1)
            f.write(np.ascontiguousarray(aX, dtype=np.float64).tobytes())
        for fam in fams:
            f.write(np.ascontiguousarray(fam["grid"], dtype=np.float64).tobytes())
    out = subprocess.run([os.path.join(binaries, "stereo_test3"), p] + (["ex"] if ex else ["test7"] if test7 else []),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    v = [float(x) for x in out.stdout.split()]
    q, t = np.array(v[:4]), np.array(v[4:7])
    loss = oracle.LOSS_TRIVIAL if (ex or test7) else oracle.LOSS_CAUCHY
    step = 3000 // 1000 if test7 else 1   # (iterStep comes from the first family's count for both loops, :2590-2611)
    O1 = oracle.OracleProblem(fams[0]["grid"], *K1, loss=loss, distortion=dist if ex else None)
    O2 = oracle.OracleProblem(fams[1]["grid"], *K2, loss=loss, distortion=dist if ex else None, T12=T12)
    qo, to, so = oracle.solve_terms([O1, O2], [fams[0]["xyz"][::step].copy(), fams[1]["xyz"][::step].copy()], [1, 0, 0, 0], [0, 0, 0],
                                    max_num_iterations=100 if (ex or test7) else 50)
    assert synth.rotation_angle_between(q, qo) < 1e-7 and np.linalg.norm(t - to) < 1e-7
    assert int(v[8]) == so["termination"]
    assert "Use Point count = %d" % (len(range(0, 3000, step)) + len(range(0, 2000, step))) in out.stderr


def test_c_abi_from_plain_c(binaries, hip, bundled_pair, tmp_path):
    """examples/c_abi_demo.c: C99, gcc, include/ea_hip.h and nothing else.  Same inputs through the C program and through
    the ctypes stub: the same bits come back."""
    X = np.ascontiguousarray(bundled_pair["aX"][:3, ::7].T)
    grid = np.ascontiguousarray(bundled_pair["grids"][3], dtype=np.float64)
    pts, gr = str(tmp_path / "points.f64"), str(tmp_path / "grid.f64")
    X.tofile(pts); grid.tofile(gr)
    K = bundled_pair["K"]
    out = subprocess.run([os.path.join(binaries, "c_abi_demo"), str(X.shape[0]), str(grid.shape[0]), str(grid.shape[1]),
                          *[repr(float(k)) for k in K], pts, gr], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    v = [float(x) for x in out.stdout.split()]
    P = hip.Problem(*K, dtype=hip.EA_F64)
    P.set_points(X); P.set_dt_grid(grid); P.set_loss(hip.LOSS_CAUCHY, 1.0)
    q, t, s = P.solve([1.0, 0, 0, 0], [0.0, 0, 0])
    P.close()
    assert v[:4] == list(q) and v[4:7] == list(t)
    assert int(v[7]) == s["num_iterations"] and int(v[8]) == s["termination"] and v[9] == s["final_cost"]
    assert 0.0 <= v[10] < 1e-12   # ea_eval_rows from C: J^T r of the rows is the gradient ea_eval reports

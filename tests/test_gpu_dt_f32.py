"""fp64 problems over a float32-STORED distance transform.

Every DT image the reference produces is CV_32F (get_distance_transform*, standalone/utils.cpp:79-82) and reaches Ceres
through cv2eigen as doubles that are exactly floats (standalone_edge_align.cpp:205-206).  The library detects that on the
device while it pads / transposes the upload (and knows it by construction for its own producers), keeps a float32
mirror of the image, and lets the plain fp64 kernels fetch a stencil row with ONE 16-byte load instead of two, widening
the four texels in registers.  The doubles the arithmetic sees are the same doubles: every result must be BIT-IDENTICAL
to the fp64-image form (tuning key "dt_f32" = 0 switches the mirror off).  Grids that are not float-representable keep
the fp64 image."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
Q0, T0 = np.array([1.0, 0, 0, 0]), np.zeros(3)


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("cost", "JtJ", "Jtr", "n_invalid"))


def _mirror(B, m=1):
    """whether the batch's kernels read the float32 mirror (resolved when the batch is next built: evaluate first)"""
    B.eval(np.tile(Q0, (m, 1)), np.zeros((m, 3)))
    return B.info("dt_f32")


def _check_batch_bit_identical(hip, B, m, q, t):
    on = B.eval(q, t)
    assert B.info("dt_f32") == 1
    r_on, J_on, bad_on = B.eval_rows(q, t, corrected=True, layout=0)
    rc_on, Jc_on, _ = B.eval_rows(q, t, corrected=False, layout=1)
    K = 4
    qk = np.stack([q] * K); tk = np.stack([t + 0.001 * k for k in range(K)])
    p_on = B.eval_poses(qk, tk)
    s_on = B.solve(q, t)
    B.set_tuning("dt_f32", 0)
    off = B.eval(q, t)
    assert B.info("dt_f32") == 0
    r_off, J_off, bad_off = B.eval_rows(q, t, corrected=True, layout=0)
    rc_off, Jc_off, _ = B.eval_rows(q, t, corrected=False, layout=1)
    p_off = B.eval_poses(qk, tk)
    s_off = B.solve(q, t)
    B.set_tuning("dt_f32", -1)
    assert _same(on, off)
    assert np.array_equal(r_on, r_off) and np.array_equal(J_on, J_off) and bad_on == bad_off
    assert np.array_equal(rc_on, rc_off) and np.array_equal(Jc_on, Jc_off)
    assert _same(p_on, p_off)
    assert np.array_equal(s_on[0], s_off[0]) and np.array_equal(s_on[1], s_off[1])
    for a, b in zip(s_on[2], s_off[2]):
        assert a["num_iterations"] == b["num_iterations"] and np.array_equal(a["it_cost"], b["it_cost"]) and a["final_cost"] == b["final_cost"]


def test_bundled_pair_fp32_image_is_bit_identical(hip, bundled_pair):
    """golden pair (frame 1 against frames 3 and 5, grids uploaded through ea_problem_set_dt): every launch shape"""
    Ps = []
    for b in (3, 5):
        P = hip.Problem(*bundled_pair["K"], dtype=hip.EA_F64)
        P.set_points(np.ascontiguousarray(bundled_pair["aX"].T[:, :3])); P.set_dt_grid(bundled_pair["grids"][b]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    B = hip.Batch(Ps)
    try:
        q = np.tile(Q0, (2, 1)); t = np.zeros((2, 3))
        for ppt in (1, 2):
            for nt in (256, 1024):
                for buf in (1, 0):
                    B.set_tuning("points_per_thread", ppt); B.set_tuning("threads", nt); B.set_tuning("buffer_loads", buf)
                    _check_batch_bit_identical(hip, B, 2, q, t)
        # a single problem through ea_eval / ea_solve (the self batch)
        e_on = Ps[0].eval(Q0, T0)
        s_on = Ps[0].solve(Q0, T0)
    finally:
        B.close()
        for P in Ps:
            P.close()


def test_device_producers_and_device_images_feed_the_mirror(hip):
    """set_now_frame writes both images (exact by construction: the transform is computed in float32);
    ea_problem_set_dt_image_device checks on the device"""
    import os
    frames = synth.load_bundled_frames(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rgbd"))
    P = hip.Problem(*synth.TUM_K, dtype=hip.EA_F64)
    P.set_ref_frame(frames[1][0], frames[1][1], z_scaling=5000.0)
    P.set_now_frame(frames[3][0])
    P.set_loss(hip.LOSS_CAUCHY, 1.0)
    B = hip.Batch([P])
    try:
        _check_batch_bit_identical(hip, B, 1, Q0[None], T0[None])
        for producer in (lambda: P.set_now_frame_canny(frames[3][0]), lambda: P.set_now_frame(frames[5][0], normalize=False)):
            producer()
            _check_batch_bit_identical(hip, B, 1, Q0[None], T0[None])
    finally:
        B.close(); P.close()


def test_grids_that_are_not_floats_keep_the_fp64_image(hip, oracle):
    base = synth.make_problem(120, 160, 4000, 40, 1, 130.0, 130.0, 79.5, 59.5, normalize=True)
    grid = base["grid"].copy()
    P = hip.Problem(*base["K"], dtype=hip.EA_F64)
    P.set_points(base["xyz"]); P.set_dt_grid(grid)
    B = hip.Batch([P])
    try:
        assert _mirror(B) == 1
        exact = B.eval(Q0[None], T0[None])
        grid[17, 23] += 1e-12                    # one texel that no float holds
        P.set_dt_grid(grid)
        assert _mirror(B) == 0
        e = oracle.OracleProblem(grid, *base["K"]).eval(base["xyz"], Q0, T0)
        got = B.eval(Q0[None], T0[None])
        assert abs(got["cost"][0] - e["cost"]) <= 1e-11 * e["cost"]
        assert not np.array_equal(got["cost"], exact["cost"]) or abs(got["cost"][0] - exact["cost"][0]) < 1e-10
        grid[17, 23] = np.nan                    # neither does a NaN survive the round trip test
        P.set_dt_grid(grid)
        assert _mirror(B) == 0
        P.set_dt_grid(base["grid"])
        assert _mirror(B) == 1 and _same(B.eval(Q0[None], T0[None]), exact)
        # variant functors and the LDS-staged form read the fp64 image
        P.set_distortion(0.01, -0.002, 0.0005, -0.0003, 0.0)
        assert _mirror(B) == 0
        P.set_distortion(0, 0, 0, 0, 0)
        B.set_tuning("use_lds", 1)
        assert _mirror(B) == 0
        B.set_tuning("use_lds", 0)
        assert _mirror(B) == 1
    finally:
        B.close(); P.close()
    P32 = hip.Problem(*base["K"], dtype=hip.EA_F32)
    P32.set_points(base["xyz"]); P32.set_dt_grid(base["grid"])
    B32 = hip.Batch([P32])
    try:
        B32.eval(Q0[None], T0[None])
        assert B32.info("dt_f32") == 0
    finally:
        B32.close(); P32.close()


def test_mixed_batches_fall_back_as_a_whole(hip):
    """a batch reads the mirror only when EVERY term has one; the results do not depend on it"""
    base = synth.make_problem(120, 160, 3000, 40, 1, 130.0, 130.0, 79.5, 59.5, normalize=True)
    noisy = base["grid"] + 1e-13 * np.random.default_rng(0).random(base["grid"].shape)
    Pa = hip.Problem(*base["K"], dtype=hip.EA_F64); Pa.set_points(base["xyz"]); Pa.set_dt_grid(base["grid"])
    Pb = hip.Problem(*base["K"], dtype=hip.EA_F64); Pb.set_points(base["xyz"][::2]); Pb.set_dt_grid(noisy)
    Ba, Bab = hip.Batch([Pa]), hip.Batch([Pa, Pb])
    try:
        alone = Ba.eval(Q0[None], T0[None])
        both = Bab.eval(np.tile(Q0, (2, 1)), np.zeros((2, 3)))
        assert Ba.info("dt_f32") == 1 and Bab.info("dt_f32") == 0
        assert alone["cost"][0] == both["cost"][0] and np.array_equal(alone["JtJ"][0], both["JtJ"][0])
    finally:
        Ba.close(); Bab.close(); Pa.close(); Pb.close()

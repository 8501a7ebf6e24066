"""ea_batch_eval_poses / ea_batch_set_poses + ea_batch_eval_resident_poses: K evaluations of every problem of a batch at K
DIFFERENT poses in one call (what K calls of ceres::Problem::Evaluate give, src/SolveEA.cpp:241), against the CPU oracle at
EACH of the K poses and against ea_batch_eval pose by pose.

The pose is a batch dimension of the launch: G poses per evaluation launch (the descriptor table replicated G times, copy g
owning the partial rows and the pose slot of pose g) + one fold launch, ceil(K / G) such pairs.

Tolerances: fp64 1e-11 / fp32 1e-4 relative against the oracle (the bars of test_gpu_shapes.py); against ea_batch_eval
1e-13 (fp64) / 1e-6 (fp32) relative -- the same partial rows in the same summation order, but the pose constants are built
on the device (make_pose_state in a kernel) instead of on the host (same formulas, fused multiply-adds placed by another
compiler); a second run of the same poses must land on the same bits."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _poses(rng, K, n, scale=1.0):
    q = np.zeros((K, n, 4)); t = np.zeros((K, n, 3))
    for k in range(K):
        for i in range(n):
            ax = rng.normal(size=3)
            q[k, i] = synth.quat_from_axis_angle(ax, np.deg2rad(scale * rng.uniform(0.0, 1.5)))
            t[k, i] = scale * rng.uniform(-0.03, 0.03, size=3)
    q[0, 0] = [1.0, 0, 0, 0]; t[0, 0] = 0.0   # (the identity is one of them)
    return q, t


@pytest.mark.parametrize("dtype_name,tol,tol_eval", [("EA_F64", 1e-11, 1e-13), ("EA_F32", 1e-4, 2e-6)])
def test_k_poses_match_the_oracle_at_every_pose(hip, oracle, dtype_name, tol, tol_eval):
    dtype = getattr(hip, dtype_name)
    rng = np.random.default_rng(23)
    base = synth.make_problem(120, 160, 9000, 40, 1, 130.0, 130.0, 79.5, 59.5,
                              planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)),
                              planted_t=(0.01, -0.005, 0.02), normalize=True)
    sizes = (9000, 257, 4097)   # ragged batch
    probs, clouds = [], []
    for n in sizes:
        X = base["xyz"][rng.choice(9000, n, replace=False)]
        P = hip.Problem(*base["K"], dtype=dtype)
        P.set_points(X); P.set_dt_grid(base["grid"]); P.set_loss(hip.LOSS_CAUCHY, 0.7)
        probs.append(P); clouds.append(X)
    O = oracle.OracleProblem(base["grid"], *base["K"], loss=hip.LOSS_CAUCHY, loss_a=0.7)
    B = hip.Batch(probs)
    try:
        for K in (1, 2, 3, 8):
            q, t = _poses(rng, K, len(probs))
            got = B.eval_poses(q, t)
            assert got["cost"].shape == (K, 3) and got["JtJ"].shape == (K, 3, 6, 6)
            for k in range(K):
                ref = B.eval(q[k], t[k])
                for i in range(len(probs)):
                    e = O.eval(clouds[i], q[k, i], t[k, i])
                    where = (dtype_name, K, k, i)
                    assert abs(got["cost"][k, i] - e["cost"]) <= tol * abs(e["cost"]), where
                    assert np.abs(got["JtJ"][k, i] - e["JtJ"]).max() <= tol * np.abs(e["JtJ"]).max(), where
                    assert np.abs(got["Jtr"][k, i] - e["Jtr"]).max() <= tol * np.abs(e["Jtr"]).max(), where
                    assert got["n_invalid"][k, i] == e["n_invalid"] == 0
                assert _rel(got["cost"][k], ref["cost"]) <= tol_eval and _rel(got["JtJ"][k], ref["JtJ"]) <= tol_eval
                assert _rel(got["Jtr"][k], ref["Jtr"]) <= tol_eval
            # resident poses: a second run of the same K poses lands on the same bits, without a new upload
            again = B.eval_resident_poses()
            assert all(np.array_equal(again[f], got[f]) for f in ("cost", "JtJ", "Jtr", "n_invalid")), K
            # ... and so does any split of the K poses over evaluation launches (G poses per launch: 1, 3, K)
            for g in (1, 3, 0):
                B.set_tuning("poses_per_launch", g)
                split = B.eval_poses(q, t)
                assert B.info("poses_per_launch") == (min(g, K) if g else K), (K, g)
                assert all(np.array_equal(split[f], got[f]) for f in ("cost", "JtJ", "Jtr", "n_invalid")), (K, g)
        # every launch shape the batch can resolve to
        q, t = _poses(rng, 5, len(probs))
        first = None
        for ppt in (1, 2, 4):
            for nt in (256, 1024):
                B.set_tuning("points_per_thread", ppt); B.set_tuning("threads", nt)
                with pytest.raises(hip.EAError) as ei:   # a change of the batch drops the resident poses
                    B.eval_resident_poses()
                assert ei.value.code == hip.EA_ERR_STATE
                got = B.eval_poses(q, t)
                first = first or got
                for f in ("cost", "JtJ", "Jtr"):
                    assert _rel(got[f], first[f]) <= (1e-12 if dtype_name == "EA_F64" else 1e-5), (ppt, nt, f)
    finally:
        B.close()
        for P in probs:
            P.close()


def test_k_poses_on_batches_the_riding_fold_does_not_cover(hip, oracle):
    """variant functors, terms sharing a pose, LDS staging: the plain evaluation + fold pair per pose, same results"""
    rng = np.random.default_rng(5)
    base = synth.make_problem(120, 160, 5000, 40, 1, 130.0, 130.0, 79.5, 59.5, normalize=True)
    P = hip.Problem(*base["K"], dtype=hip.EA_F64)
    P.set_points(base["xyz"]); P.set_dt_grid(base["grid"])
    T = hip.Problem(*base["K"], dtype=hip.EA_F64)
    T.set_points(base["xyz"][:1500]); T.set_dt_grid(base["grid"])
    B = hip.Batch([P])
    try:
        q, t = _poses(rng, 4, 1)
        plain = B.eval_poses(q, t)
        B.set_tuning("use_lds", 1)
        lds = B.eval_poses(q, t)
        for f in ("cost", "JtJ", "Jtr"):
            assert _rel(lds[f], plain[f]) <= 1e-13
        B.set_tuning("use_lds", 0)
        P.add_term(T)                      # two residual families on one pose
        both = B.eval_poses(q, t)
        for k in range(4):
            ref = B.eval(q[k], t[k])
            assert np.array_equal(both["cost"][k], ref["cost"]) or _rel(both["cost"][k], ref["cost"]) <= 1e-13
            assert _rel(both["JtJ"][k], ref["JtJ"]) <= 1e-13
        P.clear_terms()
        P.set_distortion(0.01, -0.002, 0.0005, -0.0003, 0.0)
        var = B.eval_poses(q, t)
        for k in range(4):
            ref = B.eval(q[k], t[k])
            assert _rel(var["cost"][k], ref["cost"]) <= 1e-13 and _rel(var["Jtr"][k], ref["Jtr"]) <= 1e-12
    finally:
        B.close(); P.close(); T.close()


def test_failed_functors_non_unit_quaternions_and_the_ros_flavour(hip, oracle):
    """a pose that puts points inside the z guard is counted per pose; a non-unit quaternion takes the general Jacobian;
    the ROS flavour's transposed rotation is applied by the device-side pose builder as by the host's"""
    base = synth.make_problem(120, 160, 3000, 40, 1, 130.0, 130.0, 79.5, 59.5, normalize=True)
    P = hip.Problem(*base["K"], dtype=hip.EA_F64)
    P.set_points(base["xyz"]); P.set_dt_grid(base["grid"])
    B = hip.Batch([P])
    try:
        zmean = float(np.mean(base["xyz"][:, 2]))
        q = np.array([[[1.0, 0, 0, 0]], [[1.0, 0, 0, 0]], [[1.02, 0.01, -0.02, 0.005]]])
        t = np.array([[[0.0, 0, 0]], [[0.0, 0, -zmean]], [[0.01, 0.0, 0.02]]])
        got = B.eval_poses(q, t)
        for k in range(3):
            ref = B.eval(q[k], t[k])
            assert got["n_invalid"][k, 0] == ref["n_invalid"][0]
            assert _rel(got["cost"][k], ref["cost"]) <= 1e-13 and _rel(got["JtJ"][k], ref["JtJ"]) <= 1e-12
        assert got["n_invalid"][0, 0] == 0 and got["n_invalid"][1, 0] > 0
        P.set_flavour(0.0, 0.001, 1)
        qr = np.array([[synth.quat_from_axis_angle([0.2, 1, -0.4], np.deg2rad(2.0))], [[1.0, 0, 0, 0]]])
        tr = np.array([[[0.01, -0.02, 0.015]], [[0.0, 0, 0]]])
        got = B.eval_poses(qr, tr)
        for k in range(2):
            ref = B.eval(qr[k], tr[k])
            assert _rel(got["cost"][k], ref["cost"]) <= 1e-13 and _rel(got["Jtr"][k], ref["Jtr"]) <= 1e-12
    finally:
        B.close(); P.close()


def test_empty_problems_report_zeros_at_every_pose(hip):
    """a problem without points (the reference's loop simply adds no block) has cost 0 and a zero system at every one of the
    K poses -- alone in its batch (no evaluation launch at all: the riding folds still owe their results) and next to others"""
    base = synth.make_problem(60, 80, 500, 12, 3, 65.0, 65.0, 39.5, 29.5, normalize=True)
    for dtype in (hip.EA_F64, hip.EA_F32):
        E = hip.Problem(*base["K"], dtype=dtype); E.set_points(np.zeros((0, 3))); E.set_dt_grid(base["grid"])
        F = hip.Problem(*base["K"], dtype=dtype); F.set_points(base["xyz"]); F.set_dt_grid(base["grid"])
        # (dirty the pinned result block first: a full problem leaves non-zero sums where the empty one's go next)
        Bf = hip.Batch([F])
        Bf.eval_poses(np.tile([1.0, 0, 0, 0], (6, 1, 1)), np.zeros((6, 1, 3)))
        Bf.close()
        for members in ([E], [E, F], [F, E, E]):
            B = hip.Batch(members)
            m = len(members)
            for K in (1, 2, 5):
                q = np.tile([1.0, 0, 0, 0], (K, m, 1)); t = 0.001 * np.arange(K * m * 3).reshape(K, m, 3)
                got = B.eval_poses(q, t)
                for k in range(K):
                    ref = B.eval(q[k], t[k])
                    for i, P in enumerate(members):
                        if P is E:
                            assert got["cost"][k, i] == 0.0 and not got["JtJ"][k, i].any() and not got["Jtr"][k, i].any() and got["n_invalid"][k, i] == 0
                        else:
                            assert abs(got["cost"][k, i] - ref["cost"][i]) <= 2e-6 * ref["cost"][i]
            B.close()
        E.close(); F.close()


def test_argument_checks_and_state(hip):
    base = synth.make_problem(60, 80, 500, 12, 3, 65.0, 65.0, 39.5, 29.5, normalize=True)
    P = hip.Problem(*base["K"], dtype=hip.EA_F64)
    P.set_points(base["xyz"]); P.set_dt_grid(base["grid"])
    B = hip.Batch([P])
    try:
        with pytest.raises(hip.EAError) as ei:
            B.eval_resident_poses(fetch=False)    # nothing resident yet
        assert ei.value.code == hip.EA_ERR_STATE
        L = hip.load()
        assert L.ea_batch_set_poses(B._h, 0, None, None) == hip.EA_ERR_INVALID_ARG
        q, t = np.tile([1.0, 0, 0, 0], (12, 1, 1)), np.tile([0.01, -0.02, 0.015], (12, 1, 1))   # (off the planted pose: cost > 0)
        assert L.ea_batch_eval_poses(B._h, 0, hip._dp(q), hip._dp(t), None, None, None, None) == hip.EA_ERR_INVALID_ARG
        B.set_poses(q, t)
        B.eval_resident_poses(fetch=False)       # results stay in the library; nothing handed back
        out = B.eval_resident_poses()
        assert np.all(out["cost"] == out["cost"][0]) and out["cost"][0, 0] > 0
        P.set_loss(hip.LOSS_HUBER, 0.2)          # the problem changed: resident poses are gone
        with pytest.raises(hip.EAError) as ei:
            B.eval_resident_poses()
        assert ei.value.code == hip.EA_ERR_STATE
        # growing K re-allocates; shrinking re-uses
        for K in (40, 3, 17):
            qk, tk = np.tile([1.0, 0, 0, 0], (K, 1, 1)), np.tile([0.01, -0.02, 0.015], (K, 1, 1))
            o = B.eval_poses(qk, tk)
            assert o["cost"].shape == (K, 1) and np.all(o["cost"] == o["cost"][0]) and o["cost"][0, 0] > 0
    finally:
        B.close(); P.close()


def test_many_problems_times_many_poses(hip):
    """40 small problems x 300 poses: one launch of 12 000 (problem, pose) columns -- inside the grid's y limit -- and the same
    again split over launches of 7 poses; spot-checked against ea_batch_eval"""
    base = synth.make_problem(60, 80, 900, 12, 3, 65.0, 65.0, 39.5, 29.5, planted_q=synth.quat_from_axis_angle([1, 2, 3], 0.01),
                              planted_t=(0.004, -0.002, 0.006), normalize=True)
    rng = np.random.default_rng(3)
    Ps = []
    for i in range(40):
        P = hip.Problem(*base["K"], dtype=hip.EA_F64)
        P.set_points(base["xyz"][rng.choice(900, int(rng.integers(1, 900)), replace=False)]); P.set_dt_grid(base["grid"])
        Ps.append(P)
    B = hip.Batch(Ps)
    try:
        K = 300
        q, t = _poses(rng, K, 40, scale=0.3)
        got = B.eval_poses(q, t)
        assert B.info("poses_per_launch") == K
        for k in (0, 137, 299):
            ref = B.eval(q[k], t[k])
            assert _rel(got["cost"][k], ref["cost"]) <= 1e-13 and _rel(got["JtJ"][k], ref["JtJ"]) <= 1e-12
            assert np.array_equal(got["n_invalid"][k], ref["n_invalid"])
        B.set_tuning("poses_per_launch", 7)
        split = B.eval_poses(q, t)
        assert all(np.array_equal(split[f], got[f]) for f in ("cost", "JtJ", "Jtr", "n_invalid"))
    finally:
        B.close()
        for P in Ps:
            P.close()

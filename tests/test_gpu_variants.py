"""the residual variants of standalone/utils.h — EAResidueEx (Brown-Conrady distortion,
:102-177), EAResidueSecondCam (second camera of a rigid rig, :179-292), EAResidueSecondCamEx (:295-421) — and
problems whose camera-1 and camera-2 blocks share one pose (standalone_edge_align.cpp:791-803, :3205-3218).
GPU (C-ABI) against the oracle's Jet<7> restatement of those functors; fp64 1e-11, fp32 1e-4."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
K1 = (130.0, 132.0, 79.5, 59.5)
K2 = (128.0, 129.0, 81.0, 58.0)
DIST = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)  # the D vector the reference prints (standalone_edge_align.cpp:156)
T12 = synth.rigid_4x4(synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.04), [0.11, 0.004, -0.012])
Q = synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0))
T = np.array([0.01, -0.005, 0.02])


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _gpu(hip, fam, dtype, K, distortion=None, T12m=None, loss=(1, 1.0)):
    P = hip.Problem(*K, dtype=dtype)
    P.set_points(fam["xyz"])
    P.set_dt_grid(fam["grid"])
    P.set_loss(*loss)
    if distortion is not None:
        P.set_distortion(*distortion)
    if T12m is not None:
        P.set_second_camera(T12m)
    return P


@pytest.mark.parametrize("name,dist,t12", [("Ex", DIST, None), ("SecondCam", None, T12), ("SecondCamEx", DIST, T12)])
def test_single_variant_matches_jet_functor(hip, oracle, name, dist, t12):
    fams = synth.make_stereo_problem(120, 160, 3000, 3000, 5, K1, K2, T12, Q, T, distortion=dist)
    fam, K = (fams[1], K2) if t12 is not None else (fams[0], K1)
    O = oracle.OracleProblem(fam["grid"], *K, distortion=dist, T12=t12)
    Qp = synth.quat_mul(synth.quat_from_axis_angle([0.2, -1, 0.4], 0.004), Q)  # near, not at, the planted pose
    for q, t in ((np.array([1.0, 0, 0, 0]), np.zeros(3)), (Qp, T + 0.002), (Q * 1.2, T)):  # last: non-unit q
        e = O.eval(fam["xyz"], q, t, oracle.JAC_JET, materialize=True)
        P = _gpu(hip, fam, hip.EA_F64, K, dist, t12)
        g = P.eval(q, t)
        assert g["n_invalid"] == e["n_invalid"]
        assert g["cost"] == pytest.approx(e["cost"], rel=1e-11)
        assert _rel(g["JtJ"], e["JtJ"]) < 1e-11 and _rel(g["Jtr"], e["Jtr"]) < 1e-11
        r, J = P.eval_points(q, t, corrected=False)
        assert np.abs(r - e["raw_r"]).max() < 1e-12 and _rel(J, e["raw_J"]) < 1e-11
        P.close()
        P = _gpu(hip, fam, hip.EA_F32, K, dist, t12)
        g = P.eval(q, t)
        assert g["cost"] == pytest.approx(e["cost"], rel=1e-4)
        assert _rel(g["JtJ"], e["JtJ"]) < 1e-4 and _rel(g["Jtr"], e["Jtr"]) < 1e-4
        P.close()


@pytest.mark.parametrize("dist", [None, DIST])
def test_stereo_problem_shares_one_pose(hip, oracle, dist):
    """camera-1 EAResidue[Ex] blocks + camera-2 EAResidueSecondCam[Ex] blocks in one problem"""
    fams = synth.make_stereo_problem(120, 160, 4000, 2500, 6, K1, K2, T12, Q, T, distortion=dist)
    O1 = oracle.OracleProblem(fams[0]["grid"], *K1, distortion=dist)
    O2 = oracle.OracleProblem(fams[1]["grid"], *K2, distortion=dist, T12=T12)
    P1 = _gpu(hip, fams[0], hip.EA_F64, K1, dist, None)
    P2 = _gpu(hip, fams[1], hip.EA_F64, K2, dist, T12)
    P1.add_term(P2)
    q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
    e = oracle.eval_terms([O1, O2], [fams[0]["xyz"], fams[1]["xyz"]], q0, t0, oracle.JAC_JET)
    g = P1.eval(q0, t0)
    assert g["cost"] == pytest.approx(e["cost"], rel=1e-11)
    assert _rel(g["JtJ"], e["JtJ"]) < 1e-11 and _rel(g["Jtr"], e["Jtr"]) < 1e-11
    # the sum of the two families evaluated separately
    s1 = _gpu(hip, fams[0], hip.EA_F64, K1, dist, None).eval(q0, t0) if dist is None else None
    qo, to, so = oracle.solve_terms([O1, O2], [fams[0]["xyz"], fams[1]["xyz"]], q0, t0)
    q, t, s = P1.solve(q0, t0)
    assert s["num_iterations"] == so["num_iterations"] and s["why"] == so["why"]
    assert synth.rotation_angle_between(q, qo) < 1e-7 and np.linalg.norm(t - to) < 1e-7
    assert synth.rotation_angle_between(q, Q) < 1e-4 and np.linalg.norm(t - T) < 1e-3  # and it is the planted pose
    assert s["num_point_evals"] % (4000 + 2500) == 0
    P1.close(); P2.close()


def test_variant_argument_checks(hip):
    fams = synth.make_stereo_problem(60, 80, 200, 200, 7, K1, K2, T12, Q, T)
    P = _gpu(hip, fams[0], hip.EA_F64, K1)
    bad = T12.copy(); bad[3, 0] = 0.1
    with pytest.raises(hip.EAError):
        P.set_second_camera(bad, np.linalg.inv(T12))
    with pytest.raises(hip.EAError):
        P.add_term(P)
    P32 = _gpu(hip, fams[1], hip.EA_F32, K2)
    with pytest.raises(hip.EAError):
        P.add_term(P32)  # dtype mismatch
    P.close(); P32.close()


def test_random_mixed_batches(hip, oracle):
    """Batches mixing plain, distorted, second-camera and stereo (two-term) problems of ragged sizes, both launch shapes
    of the variant kernel, both dtypes: batch evaluation and batch solve against the oracle, problem by problem."""
    rng = np.random.default_rng(77)
    for trial in range(6):
        dtype, tol = (hip.EA_F64, 1e-10) if trial % 2 == 0 else (hip.EA_F32, 2e-4)
        m = int(rng.integers(3, 7))
        Ps, Os, clouds, keep = [], [], [], []

        def gpu(fam, K, d, t12):
            P = _gpu(hip, fam, dtype, K, d, t12)
            keep.append(P)
            return P

        for i in range(m):
            kind = int(rng.integers(0, 4))  # 0 plain, 1 Ex, 2 SecondCam, 3 stereo pair with distortion
            Qp = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(0.3, 1.5)))
            Tp = rng.uniform(-0.02, 0.02, 3)
            dist = tuple(rng.uniform(-1, 1, 5) * np.array([0.2, 0.5, 0.005, 0.005, 0.5])) if kind in (1, 3) else None
            T12r = synth.rigid_4x4(synth.quat_from_axis_angle(rng.standard_normal(3), 0.04), rng.uniform(-0.1, 0.1, 3))
            fams = synth.make_stereo_problem(120, 160, int(rng.integers(200, 2500)), int(rng.integers(200, 2500)),
                                             100 * trial + i, K1, K2, T12r, Qp, Tp, distortion=dist)
            if kind in (0, 1):
                Ps.append(gpu(fams[0], K1, dist, None))
                Os.append([oracle.OracleProblem(fams[0]["grid"], *K1, distortion=dist)])
                clouds.append([fams[0]["xyz"]])
            elif kind == 2:
                Ps.append(gpu(fams[1], K2, None, T12r))
                Os.append([oracle.OracleProblem(fams[1]["grid"], *K2, T12=T12r)])
                clouds.append([fams[1]["xyz"]])
            else:
                P1, P2 = gpu(fams[0], K1, dist, None), gpu(fams[1], K2, dist, T12r)
                P1.add_term(P2)
                Ps.append(P1)
                Os.append([oracle.OracleProblem(fams[0]["grid"], *K1, distortion=dist),
                           oracle.OracleProblem(fams[1]["grid"], *K2, distortion=dist, T12=T12r)])
                clouds.append([fams[0]["xyz"], fams[1]["xyz"]])
        B = hip.Batch(Ps)
        q = np.tile([1.0, 0, 0, 0], (m, 1)) + 0.004 * rng.standard_normal((m, 4))
        q /= np.linalg.norm(q, axis=1)[:, None]
        t = 0.005 * rng.standard_normal((m, 3))
        try:
            for ppt in (1, 2):
                B.set_tuning("points_per_thread", ppt)
                g = B.eval(q, t)
                for i in range(m):
                    e = oracle.eval_terms(Os[i], clouds[i], q[i], t[i], oracle.JAC_JET)
                    assert int(g["n_invalid"][i]) == int(e["n_invalid"]), (trial, i, ppt)
                    assert abs(g["cost"][i] - e["cost"]) <= tol * abs(e["cost"]), (trial, i, ppt)
                    assert _rel(g["JtJ"][i], e["JtJ"]) <= tol and _rel(g["Jtr"][i], e["Jtr"]) <= tol, (trial, i, ppt)
            if dtype == hip.EA_F64:
                qs, ts, ss = B.solve(q, t)
                for i in range(m):
                    qo, to, so = oracle.solve_terms(Os[i], clouds[i], q[i], t[i])
                    assert ss[i]["why"] == so["why"] and ss[i]["num_iterations"] == so["num_iterations"], (trial, i)
                    assert synth.rotation_angle_between(qs[i], qo) < 1e-7 and np.linalg.norm(ts[i] - to) < 1e-7, (trial, i)
        finally:
            B.close()
            for P in keep:
                P.close()


@pytest.mark.parametrize("dtype_name", ["EA_F64", "EA_F32"])
def test_points_far_outside_the_view_have_zero_rows(hip, oracle, dtype_name):
    """A point whose distorted projection lands far outside the image (r^6 terms: pixel coordinates of 1e6 .. 1e12) reads
    the replicated border texel: Ceres' Horner spline gives that texel and derivatives of exactly 0, so the 1x6 row is
    exactly 0 whatever d(u,v)/d(x,y) is.  A tap-weight spline leaves texel * O(eps) there, times that factor (found by
    scripts/soak_variants.py: rows of O(10) in fp32, 0.1 in fp64, where the oracle has zeros)."""
    dist = (-0.05, 0.17, 0.002, -0.005, -0.55)
    fam = synth.make_stereo_problem(120, 160, 2000, 500, 7, K1, K2, T12, Q, T, distortion=dist)[0]
    rng = np.random.default_rng(3)
    z = rng.uniform(0.5, 5.0, size=200)
    wide = np.stack([z * rng.uniform(3, 40, size=200) * rng.choice([-1, 1], size=200),
                     z * rng.uniform(3, 40, size=200) * rng.choice([-1, 1], size=200), z], axis=1)
    xyz = np.concatenate([fam["xyz"], wide])
    P = hip.Problem(*K1, dtype=getattr(hip, dtype_name))
    P.set_points(xyz); P.set_dt_grid(fam["grid"]); P.set_loss(0, 1.0); P.set_distortion(*dist)
    X = P.get_points()                     # the cloud as the device holds it (fp32 mode rounds the coordinates)
    e = oracle.OracleProblem(fam["grid"], *K1, loss=0, distortion=dist).eval(X, Q, T, oracle.JAC_JET, materialize=True)
    r, J = P.eval_points(Q, T, corrected=False)
    wide_rows = np.abs(X[:, 0] / X[:, 2]) > 2.9
    dead = wide_rows & (np.abs(e["raw_J"]).max(axis=1) == 0.0)
    assert wide_rows.sum() == 200 and dead.sum() >= 150    # (a few wide points fold back into the view)
    assert np.abs(J[dead]).max() == 0.0
    tol = 1e-11 if dtype_name == "EA_F64" else 2e-4
    assert _rel(J, e["raw_J"]) < tol
    # fp32 reads the fp32 image: the border texel itself is rounded
    assert np.abs(r[dead] - e["raw_r"][dead]).max() <= (0.0 if dtype_name == "EA_F64" else 1e-6)
    g = P.eval(Q, T)
    assert _rel(g["JtJ"], e["JtJ"]) < tol
    P.close()

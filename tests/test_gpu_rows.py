"""Materialised mode (ea_batch_eval_rows / ea_batch_eval_rows_device: residual and 1x6 row of every point, the "EAResidue
batch Evaluate" view of SURVEY 8d) against the CPU oracle's Ceres-style rows (Jet<7> autodiff + plus-Jacobian +
corrector) and against the per-point kernel of the parity tests, in both layouts and both store forms.

Tolerances (file header of test_gpu_parity.py): fp64 r, J <= 1e-12; fp32 r <= 2e-5 absolute on a DT in [0, 1], J <= 2e-4
relative to max |J|.  Layouts and store forms move the same numbers: bit-identical to each other.  The sums of the rows
must reproduce the fused mode's JtJ / Jtr (1e-11 fp64, 5e-5 fp32) -- the property that holds at any size."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _small(seed=1, n=5000):
    q = synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0))
    return synth.make_problem(120, 160, n, 40, seed, 130.0, 130.0, 79.5, 59.5, planted_q=q,
                              planted_t=(0.01, -0.005, 0.02), normalize=True)


def _mk(hip, pr, dtype, loss=(1, 1.0), n=None):
    P = hip.Problem(*pr["K"], dtype=dtype)
    P.set_points(pr["xyz"] if n is None else pr["xyz"][:n])
    P.set_dt_grid(pr["grid"])
    P.set_loss(*loss)
    return P


Q = np.array([0.9990482, 0.0261769, -0.0348995, 0.0087265]); Q = Q / np.linalg.norm(Q)
T = np.array([0.03, -0.02, 0.05])


@pytest.mark.parametrize("corrected", [False, True])
@pytest.mark.parametrize("dtype_name,tol_r,tol_J", [("EA_F64", 1e-12, 1e-12), ("EA_F32", 2e-5, 2e-4)])
def test_rows_match_the_oracle_functor(hip, oracle, corrected, dtype_name, tol_r, tol_J):
    pr = _small(seed=3, n=3001)   # 11 full workgroups + a ragged one
    e = oracle.OracleProblem(pr["grid"], *pr["K"], loss=1, loss_a=1.0).eval(pr["xyz"], Q, T, oracle.JAC_JET, materialize=True)
    er, eJ = (e["r"], e["J"]) if corrected else (e["raw_r"], e["raw_J"])
    P = _mk(hip, pr, getattr(hip, dtype_name))
    B = hip.Batch([P])
    try:
        first = None
        for staged in (1, 0):
            for nt in (0, 1):
                B.set_tuning("rows_staged", staged); B.set_tuning("rows_nontemporal", nt)
                r, J, bad = B.eval_rows(Q, T, corrected=corrected, layout=0)
                assert bad == 0 and r.dtype == J.dtype == (np.float64 if dtype_name == "EA_F64" else np.float32)
                assert np.abs(r - er).max() < tol_r and _rel(J, eJ) < tol_J, (staged, nt)
                if first is None:
                    first = (r.copy(), J.copy())
                assert np.array_equal(r, first[0]) and np.array_equal(J, first[1]), (staged, nt)
                r1, J1, _ = B.eval_rows(Q, T, corrected=corrected, layout=1)
                assert J1.shape == (6, 3001) and np.array_equal(r1, first[0]) and np.array_equal(J1.T, first[1]), (staged, nt)
        # the per-point kernel of the parity tests computes the same arithmetic
        rp, Jp = P.eval_points(Q, T, corrected=corrected)
        assert np.abs(first[0] - rp).max() < tol_r * 1e-3 + 1e-15 and _rel(first[1], Jp) < tol_J * 1e-3 + 1e-15
    finally:
        B.close(); P.close()


@pytest.mark.parametrize("dtype_name,tol", [("EA_F64", 1e-11), ("EA_F32", 5e-5)])
def test_rows_of_a_ragged_batch_sum_to_the_fused_system(hip, dtype_name, tol):
    pr = _small(seed=5, n=9000)
    sizes = (1, 255, 0, 4097, 9000)   # an empty problem in the middle
    probs = [_mk(hip, pr, getattr(hip, dtype_name), loss=(1, 0.5), n=n) for n in sizes]
    B = hip.Batch(probs)
    m = len(sizes)
    qq, tt = np.tile(Q, (m, 1)), np.tile(T, (m, 1))
    tt[:, 0] += 0.001 * np.arange(m)   # every problem its own pose
    try:
        off = B.row_offsets()
        assert list(np.diff(off)) == list(sizes) and B.info("num_rows") == sum(sizes)  # (info after a build)
        r, J, bad = B.eval_rows(qq, tt, corrected=True, layout=0)
        assert bad == 0
        g = B.eval(qq, tt)
        for i, n in enumerate(sizes):
            ri, Ji = r[off[i]:off[i + 1]].astype(np.float64), J[off[i]:off[i + 1]].astype(np.float64)
            if n == 0:
                assert g["cost"][i] == 0.0
                continue
            slack = 50.0 if n <= 2 else 1.0
            assert _rel(Ji.T @ Ji, g["JtJ"][i]) < slack * tol and _rel(Ji.T @ ri, g["Jtr"][i]) < slack * tol, (i, n)
    finally:
        B.close()
        for P in probs:
            P.close()


def test_rows_into_caller_owned_device_memory(hip):
    import torch
    pr = _small(seed=6, n=6000)
    P = _mk(hip, pr, hip.EA_F32)
    B = hip.Batch([P])
    try:
        want_r, want_J, _ = B.eval_rows(Q, T, corrected=True, layout=0)
        n = int(B.row_offsets()[-1])
        r = torch.full((n + 8,), 7.0, dtype=torch.float32, device="cuda")
        J = torch.full((n + 8, 6), 7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        bad = B.eval_rows_device(Q, T, r.data_ptr(), J.data_ptr(), n + 8, corrected=True, layout=0)
        assert bad == 0
        # the wrapper also takes the tensors themselves and checks them (device, dtype, contiguity, size)
        r2, J2 = torch.zeros(n, dtype=torch.float32, device="cuda"), torch.zeros((n, 6), dtype=torch.float32, device="cuda")
        assert B.eval_rows_device(Q, T, r2, J2) == 0
        assert torch.equal(r2, r[:n]) and torch.equal(J2, J[:n])
        with pytest.raises(ValueError):
            B.eval_rows_device(Q, T, r2.double(), J2)            # wrong dtype
        with pytest.raises(ValueError):
            B.eval_rows_device(Q, T, r2.cpu(), J2)               # host tensor
        with pytest.raises(ValueError):
            B.eval_rows_device(Q, T, r2, J2.t())                 # not contiguous
        with pytest.raises(hip.EAError):
            B.eval_rows_device(Q, T, r2[: n - 3], J2)            # too few rows: the library refuses
        assert np.array_equal(r[:n].cpu().numpy(), want_r) and np.array_equal(J[:n].cpu().numpy(), want_J)
        assert float(r[n:].min()) == 7.0 and float(J[n:].min()) == 7.0   # nothing written past the rows
        # argument checks: capacity, alignment, a host pointer
        with pytest.raises(hip.EAError) as ei:
            B.eval_rows_device(Q, T, r.data_ptr(), J.data_ptr(), n - 1)
        assert ei.value.code == hip.EA_ERR_INVALID_ARG
        with pytest.raises(hip.EAError) as ei:
            B.eval_rows_device(Q, T, r.data_ptr() + 4, J.data_ptr(), n + 8)
        assert ei.value.code == hip.EA_ERR_INVALID_ARG
        with pytest.raises(hip.EAError) as ei:
            B.eval_rows_device(Q, T, r.data_ptr(), 0, n + 8)
        assert ei.value.code == hip.EA_ERR_INVALID_ARG
    finally:
        B.close(); P.close()


def test_failed_functor_rows_are_nan_and_counted(hip):
    pr = _small(seed=7, n=2000)
    xyz = pr["xyz"].copy()
    xyz[5, 2] = 0.004; xyz[700, 2] = -0.003     # |b_z| < 0.01 at the identity: the functor returns false (utils.h:70-73)
    for dtype in (hip.EA_F64, hip.EA_F32):
        P = hip.Problem(*pr["K"], dtype=dtype)
        P.set_points(xyz); P.set_dt_grid(pr["grid"])
        B = hip.Batch([P])
        try:
            for layout in (0, 1):
                r, J, bad = B.eval_rows(np.array([1.0, 0, 0, 0]), np.zeros(3), corrected=True, layout=layout)
                Jr = J if layout == 0 else J.T
                assert bad == 2
                nanrows = np.where(np.isnan(r))[0]
                assert list(nanrows) == [5, 700] and np.isnan(Jr[[5, 700]]).all()
                ok = np.ones(2000, bool); ok[[5, 700]] = False
                assert np.isfinite(Jr[ok]).all() and np.isfinite(r[ok]).all()
        finally:
            B.close(); P.close()


def test_rows_of_variant_functors_and_shared_pose_terms(hip, oracle):
    """EAResidueEx / EAResidueSecondCam[Ex] rows (standalone/utils.h:102-421) and a stereo problem whose two residual
    families share one pose: rows of the terms adjacent, each against the oracle's Jet<7> restatement of its functor."""
    K1, K2 = (130.0, 132.0, 79.5, 59.5), (128.0, 129.0, 81.0, 58.0)
    DIST = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    T12 = synth.rigid_4x4(synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.04), [0.11, 0.004, -0.012])
    Qp = synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)); Tp = np.array([0.01, -0.005, 0.02])
    fams = synth.make_stereo_problem(120, 160, 4000, 2500, 6, K1, K2, T12, Qp, Tp, distortion=DIST)
    O1 = oracle.OracleProblem(fams[0]["grid"], *K1, distortion=DIST)
    O2 = oracle.OracleProblem(fams[1]["grid"], *K2, distortion=DIST, T12=T12)
    q = synth.quat_mul(synth.quat_from_axis_angle([0.2, -1, 0.4], 0.004), Qp); t = Tp + 0.002
    e1 = O1.eval(fams[0]["xyz"], q, t, oracle.JAC_JET, materialize=True)
    e2 = O2.eval(fams[1]["xyz"], q, t, oracle.JAC_JET, materialize=True)
    for dtype, tol_r, tol_J in ((hip.EA_F64, 1e-12, 1e-11), (hip.EA_F32, 5e-5, 5e-4)):
        def mk(fam, K, t12):
            P = hip.Problem(*K, dtype=dtype)
            P.set_points(fam["xyz"]); P.set_dt_grid(fam["grid"]); P.set_loss(1, 1.0); P.set_distortion(*DIST)
            if t12 is not None:
                P.set_second_camera(t12)
            return P
        P1, P2 = mk(fams[0], K1, None), mk(fams[1], K2, T12)
        P1.add_term(P2)
        B = hip.Batch([P1])
        try:
            off = B.row_offsets()
            assert list(off) == [0, 6500]
            for layout in (0, 1):
                r, J, bad = B.eval_rows(q, t, corrected=True, layout=layout)
                Jr = J if layout == 0 else J.T
                assert bad == 0
                assert np.abs(r[:4000] - e1["r"]).max() < tol_r and np.abs(r[4000:] - e2["r"]).max() < tol_r
                assert _rel(Jr[:4000], e1["J"]) < tol_J and _rel(Jr[4000:], e2["J"]) < tol_J
            g = B.eval(q, t)
            Jd, rd = Jr.astype(np.float64), r.astype(np.float64)
            assert _rel(Jd.T @ Jd, g["JtJ"][0]) < (1e-11 if dtype == hip.EA_F64 else 1e-4)
            assert _rel(Jd.T @ rd, g["Jtr"][0]) < (1e-11 if dtype == hip.EA_F64 else 1e-4)
        finally:
            B.close(); P1.close(); P2.close()

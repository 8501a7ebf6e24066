"""ea_batch_set_tuning("wide_accumulate", 1): an fp32 evaluation sums in fp64 from the lane's sum of <= points_per_thread
products on (SURVEY section 7 step 3 asks for fp64 accumulation in fp32 mode; the default keeps a lane's and a wavefront's
sums in fp32 -- profiles/LOG.md section 10).

Properties checked: (1) at one point per lane the wide sums are fp64 sums of the same fp32 per-point products whatever
the workgroup size or the addressing form, so two launch shapes agree to fp64 reordering (1e-13 relative), and they are
not the default's numbers; (2) against the CPU oracle the fp32 bar of the other tests (1e-4 relative) holds for every
shape; (3) solves land on the default's pose to 1e-5 rad / 1e-5 m; (4) fp64 batches, variant functors and the LDS-staged
form ignore the key (info reports 0) and the pipelined bench form refuses it."""
import itertools

import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _problem(hip, base, dtype):
    P = hip.Problem(*base["K"], dtype=dtype)
    P.set_points(base["xyz"]); P.set_dt_grid(base["grid"]); P.set_loss(1, 1.0)
    return P


def test_wide_sums_do_not_depend_on_the_launch_shape_and_match_the_oracle(hip, oracle):
    base = synth.make_problem(120, 160, 20000, 40, 1, 130.0, 130.0, 79.5, 59.5,
                              planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)),
                              planted_t=(0.01, -0.005, 0.02), normalize=True)
    q = np.array([[0.9990482, 0.0261769, -0.0348995, 0.0087265]]); q /= np.linalg.norm(q)
    t = np.array([[0.03, -0.02, 0.05]])
    e = oracle.OracleProblem(base["grid"], *base["K"], loss=1, loss_a=1.0).eval(base["xyz"], q[0], t[0])
    P = _problem(hip, base, hip.EA_F32)
    B = hip.Batch([P])
    try:
        wide, narrow = {}, {}
        for ppt, nt, buf, w in itertools.product((1, 2, 4), (256, 1024), (0, 1), (0, 1)):
            B.set_tuning("points_per_thread", ppt); B.set_tuning("threads", nt); B.set_tuning("buffer_loads", buf)
            B.set_tuning("wide_accumulate", w)
            g = B.eval(q, t)
            assert B.info("wide_accumulate") == w
            for k in ("cost", "JtJ", "Jtr"):
                assert _rel(g[k][0], e[k]) <= 1e-4, (ppt, nt, buf, w, k)
            assert g["n_invalid"][0] == e["n_invalid"]
            (wide if w else narrow)[(ppt, nt, buf)] = g
        # one point per lane: the same fp32 products summed in fp64 in every shape
        ref = wide[(1, 256, 0)]
        for key in ((1, 1024, 0), (1, 256, 1), (1, 1024, 1)):
            for k in ("cost", "JtJ", "Jtr"):
                assert _rel(wide[key][k], ref[k]) <= 1e-13, (key, k)
        # (the default's sums are shape-independent at one point per lane too -- same butterfly, fp64 above it -- but they
        # are other numbers: the fp32 butterfly rounds where the fp64 one does not)
        assert any(not np.array_equal(narrow[(1, 256, 0)][k], ref[k]) for k in ("cost", "JtJ", "Jtr"))
        # and the wide sums sit at least as close to the oracle's as the default's do, up to the products' own rounding
        for k in ("cost", "JtJ", "Jtr"):
            assert _rel(ref[k][0], e[k]) <= _rel(narrow[(1, 256, 0)][k][0], e[k]) + 2e-7, k
    finally:
        B.close(); P.close()


def test_wide_solves_land_on_the_default_pose(hip):
    base = synth.make_problem(240, 320, 30000, 60, 3, 260.0, 260.0, 159.5, 119.5,
                              planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)),
                              planted_t=(0.01, -0.005, 0.02), normalize=True)
    P = _problem(hip, base, hip.EA_F32)
    B = hip.Batch([P])
    q0, t0 = np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3))
    try:
        qa, ta, sa = B.solve(q0, t0)
        B.set_tuning("wide_accumulate", 1)
        qb, tb, sb = B.solve(q0, t0)
        assert B.info("wide_accumulate") == 1
        assert sa[0]["termination"] == sb[0]["termination"] == hip.CONVERGENCE
        assert synth.rotation_angle_between(qa[0], qb[0]) < 1e-5 and np.linalg.norm(ta[0] - tb[0]) < 1e-5
    finally:
        B.close(); P.close()


def test_wide_key_is_ignored_where_it_does_not_apply(hip):
    base = synth.make_problem(120, 160, 3000, 40, 1, 130.0, 130.0, 79.5, 59.5, normalize=True)
    q, t = np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3))
    P64, P32 = _problem(hip, base, hip.EA_F64), _problem(hip, base, hip.EA_F32)
    B64, B32 = hip.Batch([P64]), hip.Batch([P32])
    try:
        want = B64.eval(q, t)
        B64.set_tuning("wide_accumulate", 1)
        got = B64.eval(q, t)
        assert B64.info("wide_accumulate") == 0
        assert all(np.array_equal(got[k], want[k]) for k in ("cost", "JtJ", "Jtr"))   # fp64: nothing to widen
        B32.set_tuning("wide_accumulate", 1); B32.set_tuning("use_lds", 1)
        B32.eval(q, t)
        assert B32.info("wide_accumulate") == 0                                       # LDS-staged form: not covered
        B32.set_tuning("use_lds", 0)
        B32.eval(q, t)
        assert B32.info("wide_accumulate") == 1
        with pytest.raises(hip.EAError) as ei:                                        # riding fold: plain fp32 sums only
            B32.bench_capture_pipelined(4)
        assert ei.value.code == hip.EA_ERR_STATE
        B32.bench_capture(4); B32.bench_steps(4)                                      # the serial graph runs it
        assert all(np.array_equal(B32.bench_result()[k], B32.eval(q, t)[k]) for k in ("cost", "JtJ", "Jtr"))
        P32.set_distortion(0.01, -0.002, 0.0005, -0.0003, 0.0)
        B32.eval(q, t)
        assert B32.info("wide_accumulate") == 0                                       # variant functor: not covered
    finally:
        B64.close(); B32.close(); P64.close(); P32.close()

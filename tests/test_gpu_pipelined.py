"""The pipelined form of a sequence of evaluations (ea_batch_bench_capture_pipelined: the fold of step k-1 rides in the
launch of evaluation k, a stand-alone fold closes the sequence) against ea_batch_eval and the CPU oracle: every launch
shape, both dtypes, single problems and ragged batches, sequences of 1 .. 7 steps.

Tolerances: the riding / closing folds sum the same partial rows as ea_batch_eval's fold in another (fixed) order, so
their result equals ea_batch_eval's to 1e-13 relative (fp64 sums of at most a few hundred rows); riding and closing folds
use the same order and must agree bit for bit; against the oracle the bars of test_gpu_shapes.py (fp64 1e-11, fp32 1e-4).
The serial graph (ea_batch_bench_capture) must reproduce ea_batch_eval bit for bit."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("cost", "JtJ", "Jtr", "n_invalid"))


def _close(a, b, tol=1e-13):
    return (_rel(a["cost"], b["cost"]) <= tol and _rel(a["JtJ"], b["JtJ"]) <= tol and _rel(a["Jtr"], b["Jtr"]) <= tol
            and np.array_equal(a["n_invalid"], b["n_invalid"]))


@pytest.mark.parametrize("dtype_name,tol", [("EA_F64", 1e-11), ("EA_F32", 1e-4)])
def test_riding_fold_matches_eval_for_every_shape(hip, oracle, dtype_name, tol):
    dtype = getattr(hip, dtype_name)
    rng = np.random.default_rng(11)
    base = synth.make_problem(120, 160, 9000, 40, 1, 130.0, 130.0, 79.5, 59.5,
                              planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)),
                              planted_t=(0.01, -0.005, 0.02), normalize=True)
    q = np.array([0.9990482, 0.0261769, -0.0348995, 0.0087265]); q /= np.linalg.norm(q)
    t = np.array([0.03, -0.02, 0.05])
    sizes = (1, 257, 4097, 9000)   # one partial row ... several rounds of the fold
    probs = []
    for n in sizes:
        P = hip.Problem(*base["K"], dtype=dtype)
        P.set_points(base["xyz"][rng.choice(9000, n, replace=False)]); P.set_dt_grid(base["grid"]); P.set_loss(1, 1.0)
        probs.append(P)
    try:
        for members in ([3], [0], [0, 1, 2, 3], [2, 0, 3]):
            B = hip.Batch([probs[i] for i in members])
            m = len(members)
            qq, tt = np.tile(q, (m, 1)), np.tile(t, (m, 1))
            try:
                for ppt in (1, 2, 4):
                    for nt in (256, 1024):
                        B.set_tuning("points_per_thread", ppt); B.set_tuning("threads", nt)
                        ref = B.eval(qq, tt)
                        for i, pi in enumerate(members):
                            xyz = probs[pi].get_points() if hasattr(probs[pi], "get_points") else None
                            if xyz is not None and i == 0 and ppt == 1 and nt == 256:
                                e = oracle.OracleProblem(base["grid"], *base["K"], loss=1, loss_a=1.0).eval(xyz, q, t)
                                slack = 100.0 if sizes[pi] <= 2 else 1.0
                                assert abs(ref["cost"][i] - e["cost"]) <= slack * tol * abs(e["cost"])
                        for steps in (1, 2, 3, 7):
                            where = (members, ppt, nt, steps)
                            B.bench_capture_pipelined(steps)
                            B.bench_steps(steps)
                            last = B.bench_result()
                            assert _close(last, ref), where
                            if steps >= 2:
                                assert _same(B.bench_result(riding=True), last), where
                            B.bench_steps(steps)   # a replay lands on the same bits
                            assert _same(B.bench_result(), last), where
                            # the same launches enqueued one by one (ea_batch_bench_steps_riding): the same bits
                            B.bench_steps(steps, riding=True)
                            assert _same(B.bench_result(), last), where + ("launch by launch",)
                            if steps >= 2:
                                assert _same(B.bench_result(riding=True), last), where + ("launch by launch",)
                        B.bench_capture(3)
                        B.bench_steps(3)
                        assert _same(B.bench_result(), ref), (members, ppt, nt, "serial graph")
            finally:
                B.close()
    finally:
        for P in probs:
            P.close()


def test_pipelined_form_refuses_what_it_does_not_cover(hip):
    base = synth.make_problem(120, 160, 3000, 40, 1, 130.0, 130.0, 79.5, 59.5, normalize=True)
    q, t = np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3))
    P = hip.Problem(*base["K"], dtype=hip.EA_F64)
    P.set_points(base["xyz"]); P.set_dt_grid(base["grid"])
    B = hip.Batch([P])
    try:
        with pytest.raises(hip.EAError) as ei:   # no poses on the device yet
            B.bench_capture_pipelined(4)
        assert ei.value.code == hip.EA_ERR_STATE
        B.eval(q, t)
        B.set_tuning("use_lds", 1)               # LDS-staged evaluation: not covered
        with pytest.raises(hip.EAError) as ei:
            B.bench_capture_pipelined(4)
        assert ei.value.code == hip.EA_ERR_STATE
        with pytest.raises(hip.EAError) as ei:
            B.bench_steps(4, riding=True)
        assert ei.value.code == hip.EA_ERR_STATE
        B.set_tuning("use_lds", 0)
        B.bench_capture_pipelined(4)
        B.bench_steps(4)
        assert _close(B.bench_result(), B.eval(q, t))
        P.set_distortion(0.01, -0.002, 0.0005, -0.0003, 0.0)   # a variant functor: not covered
        B.eval(q, t)
        with pytest.raises(hip.EAError) as ei:
            B.bench_capture_pipelined(4)
        assert ei.value.code == hip.EA_ERR_STATE
    finally:
        B.close(); P.close()

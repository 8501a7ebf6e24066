"""The SHIPPED trust-region state machine (edge_alignment_amd/csrc/ea_lm.h, the code the
device LM-step kernel runs) compiled for the host and driven by the oracle's evaluator:
its iterates must coincide with the oracle's own LM restatement."""
import ctypes as C

import numpy as np
import pytest

from edge_alignment_amd import synth


class LMOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int), ("jacobi_scaling", C.c_int), ("strategy", C.c_int)]


KT = 128


class ShimOut(C.Structure):
    _fields_ = [("x", C.c_double * 7), ("iteration", C.c_int), ("termination", C.c_int), ("why", C.c_int),
                ("num_successful", C.c_int), ("num_unsuccessful", C.c_int), ("num_evals", C.c_int),
                ("final_cost", C.c_double), ("it_cost", C.c_double * KT), ("it_radius", C.c_double * KT),
                ("it_successful", C.c_int * KT)]


CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def _opts(**kw):
    o = LMOptions(50, 1e-6, 1e-10, 1e-8, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 5, 1, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _run(shim, P, oracle, X, q0, t0, **kw):
    def cb(pose, acc, _):
        x = np.array([pose[i] for i in range(7)])
        e = P.eval(X, x[:4], x[4:])
        k = 0
        for a in range(6):
            for b in range(a, 6):
                acc[k] = e["JtJ"][a, b]; k += 1
        for a in range(6):
            acc[21 + a] = e["Jtr"][a]
        acc[27] = e["cost"]
        acc[28] = float(e["n_invalid"])
        for i in range(29, 32):
            acc[i] = 0.0
    out = ShimOut()
    o = _opts(**kw)
    q0 = np.asarray(q0, dtype=np.float64); t0 = np.asarray(t0, dtype=np.float64)
    shim.ea_lm_host_solve.argtypes = [C.POINTER(LMOptions), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, CB, C.c_void_p, C.POINTER(ShimOut)]
    rc = shim.ea_lm_host_solve(C.byref(o), q0.ctypes.data_as(C.POINTER(C.c_double)), t0.ctypes.data_as(C.POINTER(C.c_double)), 0, CB(cb), None, C.byref(out))
    assert rc == 0
    return out


@pytest.mark.parametrize("strategy", [0, 1])
def test_shipped_lm_follows_oracle_iterates_on_bundled_pair(lm_host_shim, oracle, bundled_pair, strategy):
    P = oracle.OracleProblem(bundled_pair["grids"][3], *bundled_pair["K"])
    X = bundled_pair["aX"][:3, ::30].T.copy()
    out = _run(lm_host_shim, P, oracle, X, [1, 0, 0, 0], [0, 0, 0], strategy=strategy)
    q, t, s = P.solve(X, [1, 0, 0, 0], [0, 0, 0], strategy=strategy)
    assert out.iteration == s["num_iterations"]
    assert oracle.WHY[out.why] == s["why"] and out.termination == s["termination"]
    assert out.num_successful == s["num_successful_steps"] and out.num_unsuccessful == s["num_unsuccessful_steps"]
    x = np.array(out.x[:])
    assert np.abs(x[:4] - q).max() < 1e-12 and np.abs(x[4:] - t).max() < 1e-12
    n = s["num_iterations"] + 1
    assert np.array(out.it_cost[:n]) == pytest.approx(s["it_cost"], rel=1e-12)
    assert np.array(out.it_radius[:n]) == pytest.approx(s["it_radius"], rel=1e-12)
    assert list(out.it_successful[:n]) == list(s["it_successful"])


def test_fast_path_and_general_form_agree_bit_for_bit(lm_host_shim, lm_host_shim_general, oracle):
    """lm_advance_fast (the usual iteration as one straight line) is lm_advance's own arithmetic in the same order: on the
    host, where nothing is contracted, every iterate of random solves -- accepted and rejected steps, caps, the three
    losses -- must come out the same with and without it."""
    from edge_alignment_amd import synth
    rng = np.random.default_rng(31)
    rejected = 0
    for trial in range(18):
        pr = synth.make_problem(60, 80, int(rng.integers(200, 900)), 12, 700 + trial, 65.0, 65.0, 39.5, 29.5,
                                planted_q=synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(0.3, 2.0))),
                                planted_t=tuple(rng.uniform(-0.03, 0.03, 3)), normalize=bool(trial % 2))
        loss = [(0, 1.0), (1, 1.0), (1, 0.2), (2, 0.3)][trial % 4]
        P = oracle.OracleProblem(pr["grid"], *pr["K"], loss=loss[0], loss_a=loss[1])
        kw = dict(max_num_iterations=int(rng.choice([4, 30])), strategy=1 if trial % 3 == 1 else 0)   # (LM and traditional dogleg)
        if trial % 2:
            kw["min_relative_decrease"] = 0.97
        if trial % 3 == 2:
            q0, t0 = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(12.0)), rng.uniform(-0.2, 0.2, 3)
            kw["initial_trust_region_radius"] = 1e8
        else:
            q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
        a = _run(lm_host_shim, P, oracle, pr["xyz"], q0, t0, **kw)
        b = _run(lm_host_shim_general, P, oracle, pr["xyz"], q0, t0, **kw)
        assert list(a.x[:]) == list(b.x[:]) and a.iteration == b.iteration and a.why == b.why and a.termination == b.termination, trial
        assert a.num_successful == b.num_successful and a.num_unsuccessful == b.num_unsuccessful and a.num_evals == b.num_evals
        assert a.final_cost == b.final_cost
        n = a.iteration + 1
        assert list(a.it_cost[:n]) == list(b.it_cost[:n]) and list(a.it_radius[:n]) == list(b.it_radius[:n])
        assert list(a.it_successful[:n]) == list(b.it_successful[:n])
        rejected += int(a.num_unsuccessful > 0)
    assert rejected >= 1


def test_shipped_lm_failure_and_limits(lm_host_shim, oracle):
    pr = synth.make_problem(120, 160, 600, 40, 21, 130.0, 130.0, 79.5, 59.5,
                            planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)), planted_t=(0.01, -0.005, 0.02), normalize=True)
    P = oracle.OracleProblem(pr["grid"], *pr["K"])
    X = pr["xyz"].copy()
    out = _run(lm_host_shim, P, oracle, X, [1, 0, 0, 0], [0, 0, 0], max_num_iterations=3)
    q, t, s = P.solve(X, [1, 0, 0, 0], [0, 0, 0], max_num_iterations=3)
    assert oracle.WHY[out.why] == s["why"] == "max_iterations" and out.iteration == 3
    assert np.abs(np.array(out.x[:4]) - q).max() < 1e-13
    X[5] = [0.0, 0.0, 0.001]  # functor returns false -> FAILURE at iteration 0, pose untouched
    out = _run(lm_host_shim, P, oracle, X, [1, 0, 0, 0], [0, 0, 0])
    assert out.termination == 2 and oracle.WHY[out.why] == "initial_eval_failed" and out.num_evals == 1
    assert list(out.x[:]) == [1, 0, 0, 0, 0, 0, 0]


def test_a_non_finite_system_fails_the_evaluation(lm_host_shim, oracle):
    """a finite cost with an Inf / NaN in the JtJ / Jtr slots (fp32 products can overflow where the residual
    does not) must not reach the factorisation.  Ceres fails such an evaluation (IsArrayValid): FAILURE at the start
    point with the pose untouched; afterwards a rejected step -- the solve goes on from the last good pose and never
    publishes a non-finite one."""
    pr = synth.make_problem(120, 160, 600, 40, 21, 130.0, 130.0, 79.5, 59.5,
                            planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)), planted_t=(0.01, -0.005, 0.02), normalize=True)
    P = oracle.OracleProblem(pr["grid"], *pr["K"])
    X = pr["xyz"].copy()

    def run(poison_eval, slot, value):
        calls = [0]
        poses = []

        def cb(pose, acc, _):
            x = np.array([pose[i] for i in range(7)])
            poses.append(x)
            e = P.eval(X, x[:4], x[4:])
            k = 0
            for a in range(6):
                for b in range(a, 6):
                    acc[k] = e["JtJ"][a, b]; k += 1
            for a in range(6):
                acc[21 + a] = e["Jtr"][a]
            acc[27] = e["cost"]; acc[28] = 0.0
            for i in range(29, 32):
                acc[i] = 0.0
            if calls[0] == poison_eval:
                acc[slot] = value
            calls[0] += 1
        out = ShimOut()
        o = _opts()
        q0 = np.array([1.0, 0, 0, 0]); t0 = np.zeros(3)
        lm_host_shim.ea_lm_host_solve.argtypes = [C.POINTER(LMOptions), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, CB, C.c_void_p, C.POINTER(ShimOut)]
        assert lm_host_shim.ea_lm_host_solve(C.byref(o), q0.ctypes.data_as(C.POINTER(C.c_double)), t0.ctypes.data_as(C.POINTER(C.c_double)), 0, CB(cb), None, C.byref(out)) == 0
        return out, poses

    clean, _ = run(-1, 0, 0.0)
    for slot, value in ((3, np.inf), (24, np.nan), (0, -np.inf), (27, np.nan)):
        out, poses = run(0, slot, value)                      # at the start point
        assert out.termination == 2 and oracle.WHY[out.why] == "initial_eval_failed" and out.num_evals == 1, (slot, value)
        assert list(out.x[:]) == [1, 0, 0, 0, 0, 0, 0]
        out, poses = run(2, slot, value)                      # at a candidate: a rejected step, then the solve goes on
        assert out.termination == 0 and out.num_unsuccessful >= 1, (slot, value)
        assert all(np.all(np.isfinite(x)) for x in poses) and np.all(np.isfinite(out.x[:]))
        assert out.it_successful[2] == 0
        assert synth.rotation_angle_between(np.array(out.x[:4]), np.array(clean.x[:4])) < 1e-6
        assert np.abs(np.array(out.x[4:]) - np.array(clean.x[4:])).max() < 1e-6


def test_pose_state_general_derivative_matches_jet(lm_host_shim, oracle):
    # G_j of make_pose_state (used by the kernels for |q| != 1) against Jet autodiff
    rng = np.random.default_rng(5)
    g = rng.random((64, 48))
    P = oracle.OracleProblem(g, 50.0, 55.0, 31.5, 23.5, loss=oracle.LOSS_TRIVIAL)
    q = np.array([0.7, 0.3, -0.4, 0.2]) * 1.2
    t = np.array([0.02, -0.01, 0.03])
    x = np.concatenate([q, t])
    R = np.zeros(9); G = np.zeros(27); u = C.c_int()
    dp = C.POINTER(C.c_double)
    lm_host_shim.ea_lm_host_pose_state(x.ctypes.data_as(dp), 0, R.ctypes.data_as(dp), G.ctypes.data_as(dp), C.byref(u))
    assert u.value == 0
    X = np.array([0.3, -0.2, 2.0])
    ok, r, jq, jt = P.block_jet(q, t, X)
    Pm = oracle.quat_plus_jacobian(q)
    j_delta = jq @ Pm
    G = G.reshape(3, 3, 3)
    got = np.array([jt @ (G[j] @ X) for j in range(3)])
    assert got == pytest.approx(j_delta, rel=1e-12, abs=1e-12)
    x[:4] = q / np.linalg.norm(q)
    lm_host_shim.ea_lm_host_pose_state(x.ctypes.data_as(dp), 0, R.ctypes.data_as(dp), G.ctypes.data_as(dp), C.byref(u))
    assert u.value == 1


def test_solve_wait_has_a_deadline(lm_host_shim):
    """ea_batch_solve / ea_solve_sharded_device poll pinned flags that a step kernel lowers; the wait is bounded
    (ea_options.solve_timeout_ms, edge_alignment_amd/csrc/ea_spin.h): a flag that is never lowered costs an error after
    the budget instead of a hung process, a flag lowered in time ends the wait at once, a negative budget waits on."""
    import ctypes as C
    import threading
    import time
    fn = lm_host_shim.ea_test_spin_until_zero
    fn.argtypes = [C.POINTER(C.c_int), C.c_double]
    flag = C.c_int(1)
    t0 = time.perf_counter()
    assert fn(C.byref(flag), 60.0) == 1          # the stubbed never-finishing flag
    el = time.perf_counter() - t0
    assert 0.05 <= el < 2.0, el
    flag.value = 1
    threading.Timer(0.05, lambda: setattr(flag, "value", 0)).start()
    t0 = time.perf_counter()
    assert fn(C.byref(flag), 5000.0) == 0        # lowered in time
    assert time.perf_counter() - t0 < 2.0
    flag.value = 1
    threading.Timer(0.15, lambda: setattr(flag, "value", 0)).start()
    assert fn(C.byref(flag), -1.0) == 0          # no deadline: waits for the flag however long it takes

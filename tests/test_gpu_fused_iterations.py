"""One launch per LM iteration (ea_lm_iter_kernel: every workgroup of the evaluation takes the LM step itself) against the
(evaluate, step) pairs it replaces: the same iterates BIT FOR BIT -- same fold order, same state machine, same chunks --
on the bundled pair, on random small problems (accepted and rejected steps, the three losses, both dtypes, iteration caps,
a non-unit start quaternion, the transposed-rotation flavour), on small batches, and when a solve is repeated on the same
handles, LM and dogleg.  Solves that do not qualify (1024-thread workgroups, more workgroups than CUs) must take the pair
form and say so.  The pair form
itself is checked against the oracle in test_gpu_lm_random.py / test_gpu_parity.py."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu

TRACE = ("it_cost", "it_cost_change", "it_gradient_max_norm", "it_step_norm", "it_relative_decrease", "it_radius", "it_successful")


def _both(hip, problems, q0, t0, **opts):
    """the same solve in both forms on one batch -> (fused result, pair result, fused flag of the first)"""
    B = hip.Batch(problems)
    out = []
    for fused in (1, 0):
        B.set_tuning("fused_iterations", -1 if fused else 0)
        q, t, s = B.solve(q0, t0, **opts)
        out.append((q, t, s, B.info("fused_iterations")))
    B.close()
    return out


def _same(a, b, where):
    qa, ta, sa, _ = a
    qb, tb, sb, _ = b
    assert np.array_equal(qa, qb) and np.array_equal(ta, tb), where
    for x, y in zip(sa, sb):
        for k in ("termination", "why", "num_iterations", "num_successful_steps", "num_unsuccessful_steps", "initial_cost",
                  "final_cost", "num_point_evals"):
            assert x[k] == y[k], (where, k, x[k], y[k])
        for k in TRACE:
            assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])), (where, k)


def _problem(hip, pr, dtype, loss, flavour=None):
    P = hip.Problem(*pr["K"], dtype=dtype)
    P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"]); P.set_loss(*loss)
    if flavour is not None:
        P.set_flavour(*flavour)
    return P


def test_bundled_pair_both_dtypes(hip, bundled_pair):
    for dtype in (hip.EA_F64, hip.EA_F32):
        for stride in (30, 1):
            P = hip.Problem(*bundled_pair["K"], dtype=dtype)
            P.set_points(np.ascontiguousarray(bundled_pair["aX"][:3, ::stride].T)); P.set_dt_grid(bundled_pair["grids"][3]); P.set_loss(1, 1.0)
            f, u = _both(hip, [P], [1.0, 0, 0, 0], [0.0, 0, 0])
            assert f[3] == 1 and u[3] == 0
            assert f[2][0]["num_iterations"] >= 5
            _same(f, u, (dtype, stride))
            P.close()


def test_random_problems_bit_identical(hip):
    rng = np.random.default_rng(77)
    rejected = capped = 0
    seen = []
    for trial in range(30):
        q_pl = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(0.2, 3.0)))
        t_pl = tuple(rng.uniform(-0.04, 0.04, 3))
        n = int(rng.integers(300, 30000))
        pr = synth.make_problem(120, 160, n, int(rng.integers(10, 60)), 3000 + trial, 130.0, 130.0, 79.5, 59.5,
                                planted_q=q_pl, planted_t=t_pl, normalize=bool(trial % 2), pixel_centres=bool(trial % 3))
        loss = [(0, 1.0), (1, 1.0), (1, 0.2), (2, 0.3)][trial % 4]
        dtype = hip.EA_F32 if trial % 3 == 1 else hip.EA_F64
        opts = dict(max_num_iterations=int(rng.choice([3, 25])),
                    strategy=hip.STRATEGY_DOGLEG if trial % 3 == 2 else hip.STRATEGY_LM)
        if trial % 2 == 1:
            opts["min_relative_decrease"] = 0.97
        if trial % 5 == 4:
            q0 = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(8.0, 25.0)))
            t0 = rng.uniform(-0.4, 0.4, 3)
            opts["initial_trust_region_radius"] = float(rng.choice([1e4, 1e8, 1e16]))
        elif trial % 5 == 3:
            q0 = np.array([1.02, 0.01, -0.02, 0.005])  # non-unit: the general-quaternion Jacobian (G through the LDS pose)
            t0 = np.zeros(3)
        else:
            q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
        flavour = (0.0, 0.001, 1) if trial % 7 == 6 else None  # ROS flavour: no guard, z + 0.001, rotation applied transposed
        P = _problem(hip, pr, dtype, loss, flavour)
        f, u = _both(hip, [P], q0, t0, **opts)
        P.close()
        assert f[3] == 1 and u[3] == 0, trial
        _same(f, u, (trial, n, loss, dtype, opts))
        rejected += int(f[2][0]["num_unsuccessful_steps"] > 0)
        capped += int(f[2][0]["why"] == "max_iterations")
        seen.append((opts["max_num_iterations"], f[2][0]["num_iterations"], f[2][0]["why"], f[2][0]["termination"]))
    assert rejected >= 4 and capped >= 2, seen  # rejected steps (cold system copied forward) and iteration caps both occurred


def test_small_batches_and_repeated_solves(hip):
    rng = np.random.default_rng(5)
    probs, q0, t0 = [], [], []
    for i in range(4):
        pr = synth.make_problem(120, 160, 2000 + 1500 * i, 30, 400 + i, 130.0, 130.0, 79.5, 59.5,
                                planted_q=synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(1.0 + i)),
                                planted_t=tuple(rng.uniform(-0.03, 0.03, 3)))
        probs.append(_problem(hip, pr, hip.EA_F64, (1, 1.0)))
        q0.append([1.0, 0, 0, 0]); t0.append([0.0, 0, 0])
    # problems of a batch end at different iterations: the finished ones sit out the later launches
    f, u = _both(hip, probs, q0, t0, max_num_iterations=30)
    assert f[3] == 1 and u[3] == 0
    assert len({s["num_iterations"] for s in f[2]}) > 1
    _same(f, u, "batch of 4")
    # a cap that cuts some problems short while others have finished: states come back from the device
    f, u = _both(hip, probs, q0, t0, max_num_iterations=6)
    _same(f, u, "batch of 4, capped")
    # a problem without a single point among them: its writer workgroup alone steps it (zero system: converged at once)
    E = hip.Problem(*synth.make_problem(120, 160, 500, 30, 77, 130.0, 130.0, 79.5, 59.5)["K"], dtype=hip.EA_F64)
    pr = synth.make_problem(120, 160, 500, 30, 77, 130.0, 130.0, 79.5, 59.5)
    E.set_points(pr["xyz"][:0]); E.set_dt_grid(pr["grid"]); E.set_loss(1, 1.0)
    f, u = _both(hip, [probs[0], E, probs[1]], q0[:3], t0[:3])
    assert f[3] == 1 and f[2][1]["num_iterations"] == 0 and f[2][1]["termination"] == 0
    _same(f, u, "batch with an empty problem")
    E.close()
    # the same handles again, and again (buffers of either parity hold what the last solve left)
    B = hip.Batch(probs[:1])
    first = B.solve(q0[0], t0[0])
    for _ in range(3):
        again = B.solve(q0[0], t0[0])
        assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])
        assert np.array_equal(first[2][0]["it_cost"], again[2][0]["it_cost"])
    assert B.info("fused_iterations") == 1
    B.close()
    for P in probs:
        P.close()


def test_solves_that_do_not_qualify_take_the_pairs(hip):
    pr = synth.make_problem(120, 160, 3000, 30, 9, 130.0, 130.0, 79.5, 59.5)
    P = _problem(hip, pr, hip.EA_F64, (1, 1.0))
    B = hip.Batch([P])
    B.solve([1.0, 0, 0, 0], [0.0, 0, 0], strategy=hip.STRATEGY_DOGLEG)
    assert B.info("fused_iterations") == 1
    B.solve([1.0, 0, 0, 0], [0.0, 0, 0])
    assert B.info("fused_iterations") == 1
    B.set_tuning("threads", 1024)  # 1024-thread workgroups: the step's registers do not fit beside them
    B.solve([1.0, 0, 0, 0], [0.0, 0, 0])
    assert B.info("fused_iterations") == 0
    B.close(); P.close()
    # more workgroups than CUs: 7e4 points take two points per lane (274 chunks otherwise) and still qualify;
    # forced to one point per lane they do not
    pr = synth.make_problem(240, 320, 70000, 60, 10, 260.0, 260.0, 159.5, 119.5)
    P = _problem(hip, pr, hip.EA_F64, (1, 1.0))
    B = hip.Batch([P])
    B.solve([1.0, 0, 0, 0], [0.0, 0, 0], max_num_iterations=3)
    assert B.info("fused_iterations") == 1 and B.info("points_per_thread") == 2
    B.set_tuning("points_per_thread", 1)
    B.solve([1.0, 0, 0, 0], [0.0, 0, 0], max_num_iterations=3)
    assert B.info("fused_iterations") == 0
    B.close(); P.close()
    # two problems of 1.4e5 fp32 points: 2 x 274 chunks
    pr = synth.make_problem(240, 320, 70000, 60, 11, 260.0, 260.0, 159.5, 119.5)
    Ps = [_problem(hip, pr, hip.EA_F32, (1, 1.0)) for _ in range(2)]
    B = hip.Batch(Ps)
    B.solve([[1.0, 0, 0, 0]] * 2, [[0.0, 0, 0]] * 2, max_num_iterations=3)
    assert B.info("fused_iterations") == 0
    B.close()
    for P in Ps:
        P.close()


def test_start_that_cannot_be_evaluated(hip):
    """a functor that returns false at the start pose: FAILURE at iteration 0, parameters untouched, in both forms"""
    pr = synth.make_problem(120, 160, 2000, 30, 11, 130.0, 130.0, 79.5, 59.5)
    xyz = pr["xyz"].copy()
    xyz[5, 2] = 0.001  # inside the (-0.01, 0.01) guard
    pr["xyz"] = xyz
    P = _problem(hip, pr, hip.EA_F64, (1, 1.0))
    f, u = _both(hip, [P], [1.0, 0, 0, 0], [0.0, 0, 0])
    assert f[3] == 1
    assert f[2][0]["termination"] == 2 and np.array_equal(f[0][0], [1.0, 0, 0, 0])
    _same(f, u, "invalid start")
    P.close()

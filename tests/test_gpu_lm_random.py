"""Device trust-region solves against the CPU oracle's loop on random small problems: random planted poses, starts from
identity / near the planted pose / far away (radius 1e4 / 1e8 / 1e16), an acceptance threshold of 0.97 in half of the
solves (rejected steps, shrinking radius), the three losses, LM and dogleg, iteration caps of 5 / 25.  Same termination, same accept / reject pattern, cost trace to 1e-6 (while the cost
is above 1e-9 of the initial one: below that a trivial-loss cost is rounding noise of its own sum)."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu


def test_random_solves_follow_the_oracle(hip, oracle):
    rng = np.random.default_rng(2024)
    rejected_seen = 0
    for trial in range(36):
        q_pl = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(0.2, 3.0)))
        t_pl = tuple(rng.uniform(-0.04, 0.04, 3))
        n = int(rng.integers(300, 5000))
        pr = synth.make_problem(120, 160, n, int(rng.integers(10, 60)), 1000 + trial, 130.0, 130.0, 79.5, 59.5,
                                planted_q=q_pl, planted_t=t_pl, normalize=bool(trial % 2), pixel_centres=bool(trial % 3))
        loss = [(0, 1.0), (1, 1.0), (1, 0.2), (2, 0.3)][trial % 4]
        opts = dict(strategy=hip.STRATEGY_DOGLEG if trial % 3 == 2 else hip.STRATEGY_LM,
                    max_num_iterations=int(rng.choice([5, 25])))
        if trial % 2 == 1:
            opts["min_relative_decrease"] = 0.97  # most steps of a far start fall short of it: reject, shrink, retry
        mode = trial % 5 if trial % 2 else 4
        if mode == 0:
            q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
        elif mode < 4:
            q0 = q_pl + 0.01 * rng.standard_normal(4)
            q0 /= np.linalg.norm(q0)
            t0 = np.asarray(t_pl) + 0.01 * rng.standard_normal(3)
        else:
            q0 = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(8.0, 25.0)))
            t0 = rng.uniform(-0.4, 0.4, 3)
            opts["initial_trust_region_radius"] = float(rng.choice([1e4, 1e8, 1e16]))
        qo, to, so = oracle.OracleProblem(pr["grid"], *pr["K"], loss=loss[0], loss_a=loss[1]).solve(pr["xyz"], q0, t0, **opts)
        P = hip.Problem(*pr["K"], dtype=hip.EA_F64)
        P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"]); P.set_loss(*loss)
        q, t, s = P.solve(q0, t0, **opts)
        P.close()
        rejected_seen += int(so["num_unsuccessful_steps"] > 0)
        where = (trial, n, loss, opts)
        assert s["why"] == so["why"] and s["num_iterations"] == so["num_iterations"], where
        assert list(s["it_successful"]) == list(so["it_successful"]), where
        live = so["it_cost"] > 1e-9 * so["initial_cost"]
        assert np.allclose(s["it_cost"][live], so["it_cost"][live], rtol=1e-6, atol=0), where
        if live.all():
            assert synth.rotation_angle_between(q, qo) < 1e-6 and np.linalg.norm(t - to) < 1e-6, where
    assert rejected_seen >= 5  # the sample does exercise the reject / shrink branch

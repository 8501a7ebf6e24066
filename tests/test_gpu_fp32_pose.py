"""north_star's pose bar for the fp32 path: on random planted-pose problems (VGA and QVGA images, 3e3-4e4 points, the three
losses, normalised and raw DT, pixel-centre and sub-pixel points) the fp32 device solve and the fp64 oracle solve from
the identity converge to poses within 1e-4 rad / 1e-3 m of each other (measured worst case: 8e-7 rad, 3e-6 m)."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu


def test_fp32_solves_meet_the_pose_bar(hip, oracle):
    rng = np.random.default_rng(31)
    converged = 0
    for trial in range(14):
        q_pl = synth.quat_from_axis_angle(rng.standard_normal(3), np.deg2rad(rng.uniform(0.2, 2.0)))
        t_pl = tuple(rng.uniform(-0.03, 0.03, 3))
        H, W = (240, 320) if trial % 2 else (480, 640)
        n = int(rng.integers(3000, 40000))
        pr = synth.make_problem(H, W, n, int(rng.integers(60, 300)), 4000 + trial, 0.8 * W, 0.8 * W, W / 2 - 0.5, H / 2 - 0.5,
                                planted_q=q_pl, planted_t=t_pl, normalize=bool(trial % 3), pixel_centres=bool(trial % 2))
        loss = [(1, 1.0), (0, 1.0), (2, 0.3)][trial % 3]
        q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
        qo, to, so = oracle.OracleProblem(pr["grid"], *pr["K"], loss=loss[0], loss_a=loss[1]).solve(pr["xyz"], q0, t0)
        P = hip.Problem(*pr["K"], dtype=hip.EA_F32)
        P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"]); P.set_loss(*loss)
        q, t, s = P.solve(q0, t0)
        P.close()
        if so["termination"] == 0 and s["termination"] == 0:  # both CONVERGENCE
            converged += 1
            assert synth.rotation_angle_between(q, qo) < 1e-4 and np.linalg.norm(t - to) < 1e-3, (trial, n, loss)
    assert converged >= 10

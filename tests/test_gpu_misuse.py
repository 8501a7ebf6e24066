"""C-ABI misuse (SURVEY 8b "Errors": status codes, no exceptions, nothing undefined across the boundary): every entry point
with a live handle and one bad argument -- NULL data, zero / negative / overflowing extents, unknown enum values, NaN
thresholds and intrinsics, invalid ceres::Solver::Options (Options::IsValid), stale members.  Each call must return an
error code (or behave as documented) and leave the handle evaluating the reference problem to the same bits.  The cases
live in scripts/misuse_probe.py, which also runs each of them in a forked child to report crashes (none)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe():
    spec = importlib.util.spec_from_file_location("misuse_probe", os.path.join(ROOT, "scripts", "misuse_probe.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_misuse_case_is_refused_and_leaves_the_handle_intact(hip):
    res = _probe().run_in_process()
    assert len(res) >= 48
    bad = [(v, n, m) for v, n, m in res if v != "ok" and "1x1 grid" not in n]
    assert not bad, "\n".join("%s: %s -- %s" % b for b in bad)
    # a 1x1 distance-transform grid is legal (Grid2D clamps every tap to its one texel): accepted, handle intact
    assert [v for v, n, m in res if "1x1 grid" in n] == ["ACCEPTED"]


def test_non_finite_start_is_a_failed_initial_evaluation(hip):
    """ceres: a residual block producing non-finite values fails the evaluation (ResidualBlock::Evaluate -> IsArrayValid);
    at the start point the solve ends with FAILURE and the parameters untouched -- not CONVERGENCE at a NaN cost."""
    from edge_alignment_amd import synth
    pr = synth.make_problem(96, 128, 2000, 12, 5, 120.0, 121.0, 63.5, 47.5, normalize=True)
    for dtype in (hip.EA_F64, hip.EA_F32):
        P = hip.Problem(*pr["K"], dtype=dtype)
        P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"])
        q0 = np.array([np.nan, 0, 0, 0.0])
        q, t, s = P.solve(q0, np.zeros(3))
        assert s["termination"] == hip.FAILURE and s["why"] == "initial_eval_failed" and s["num_iterations"] == 0
        assert np.isnan(q[0]) and np.array_equal(q[1:], q0[1:]) and not t.any()
        X = pr["xyz"].copy(); X[7, 0] = np.inf       # one non-finite point
        P.set_points(X)
        q, t, s = P.solve([1, 0, 0, 0], np.zeros(3))
        assert s["termination"] == hip.FAILURE and s["why"] == "initial_eval_failed"
        P.close()


def test_invalid_solver_options_are_refused(hip):
    from edge_alignment_amd import synth
    pr = synth.make_problem(96, 128, 500, 12, 5, 120.0, 121.0, 63.5, 47.5, normalize=True)
    P = hip.Problem(*pr["K"], dtype=hip.EA_F64)
    P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"])
    for kw in (dict(max_num_iterations=-1), dict(strategy=9), dict(initial_trust_region_radius=-1.0), dict(initial_trust_region_radius=float("nan")),
               dict(function_tolerance=float("nan")), dict(function_tolerance=-1.0), dict(min_trust_region_radius=0.0),
               dict(max_trust_region_radius=1.0), dict(min_lm_diagonal=2.0, max_lm_diagonal=1.0), dict(min_relative_decrease=-0.5),
               dict(iterations_per_sync=-1), dict(solve_timeout_ms=float("nan"))):
        with pytest.raises(hip.EAError) as ei:
            P.solve([1, 0, 0, 0], [0, 0, 0], **kw)
        assert ei.value.code == -1, kw
    q, t, s = P.solve([1, 0, 0, 0], [0, 0, 0], max_num_iterations=0)     # legal: evaluate, take no step
    assert s["num_iterations"] == 0 and s["termination"] == hip.NO_CONVERGENCE
    P.close()

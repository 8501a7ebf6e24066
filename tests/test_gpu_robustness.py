"""Error paths of the device solve that need a GPU: the deadline of the host's wait, a failed initial evaluation, the
tracker's all-or-nothing push."""
import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
Q0, T0 = np.array([1.0, 0, 0, 0]), np.zeros(3)


def _problem(hip, n=4000, dtype=None):
    pr = synth.make_problem(120, 160, n, 40, 1, 130.0, 130.0, 79.5, 59.5,
                            planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)),
                            planted_t=(0.01, -0.005, 0.02), normalize=True)
    P = hip.Problem(*pr["K"], dtype=dtype if dtype is not None else hip.EA_F64)
    P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"])
    return P, pr


def test_solve_deadline_returns_an_error_and_the_batch_recovers(hip):
    """A stream that makes no progress (held on a host function here; a kernel that never lowers its flag in the
    field) trips ea_options.solve_timeout_ms: EA_ERR_HIP with a message, not a hung process.  The batch drains the
    abandoned launches before its next use and then solves as if nothing had happened."""
    P, pr = _problem(hip)
    B = hip.Batch([P])
    q_ref, t_ref, s_ref = B.solve(Q0, T0)
    B.set_tuning("test_stall_ms", 300)
    with pytest.raises(hip.EAError) as ei:
        B.solve(Q0, T0, solve_timeout_ms=30.0)
    assert ei.value.code == hip.EA_ERR_HIP and "deadline" in str(ei.value)
    B.set_tuning("test_stall_ms", 0)
    q, t, s = B.solve(Q0, T0)                       # drains the stalled launches first
    assert np.array_equal(q, q_ref) and np.array_equal(t, t_ref)
    assert s[0]["num_iterations"] == s_ref[0]["num_iterations"]
    B.set_tuning("test_stall_ms", 40)
    q, t, s = B.solve(Q0, T0, solve_timeout_ms=-1.0)  # no deadline: waits the stall out
    assert np.array_equal(q, q_ref)
    B.close(); P.close()


def test_failed_initial_evaluation_reports_defined_costs(hip):
    """utils.h:70-73: a block whose point lands inside |z| < 0.01 returns false; at the start pose Ceres fails the solve.
    No trace row exists then: the summary carries Ceres' -1 costs, not whatever the trace buffer held before."""
    P, pr = _problem(hip)
    q_ok, t_ok, s_ok = P.solve(Q0, T0)              # leaves a full trace behind in the batch's buffers
    assert s_ok["termination"] == hip.CONVERGENCE
    xyz = pr["xyz"].copy()
    xyz[7] = [0.1, 0.2, 0.001]                      # b_z inside the guard band at the identity pose
    P.set_points(xyz)
    q, t, s = P.solve(Q0, T0)
    assert s["termination"] == hip.FAILURE and s["why"] == "initial_eval_failed"
    assert s["initial_cost"] == -1.0 and s["final_cost"] == -1.0 and s["num_iterations"] == 0
    assert np.array_equal(q, Q0) and np.array_equal(t, T0)
    P.close()


def test_tracker_push_is_all_or_nothing(hip):
    """A push rejected for its arguments leaves the tracker exactly as it was (round 1 checked z_scaling after it had
    replaced the DT image and advanced the prior)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fr = synth.load_bundled_frames(os.path.join(root, "tests", "golden", "rgbd"))
    TA, TB = hip.Tracker(*synth.TUM_K), hip.Tracker(*synth.TUM_K)
    for k in (1, 2):
        ra = TA.push_frame(fr[k][0], fr[k][1])
        rb = TB.push_frame(fr[k][0], fr[k][1])
    with pytest.raises(hip.EAError):
        TA.push_frame(fr[3][0], fr[3][1], z_scaling=0.0)      # rejected before anything is touched
    ra = TA.push_frame(fr[3][0], fr[3][1])
    rb = TB.push_frame(fr[3][0], fr[3][1])
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
    TA.close(); TB.close()


def test_failed_descriptor_build_leaves_the_batch_dirty(hip):
    """A build that fails after the problems changed (a failed allocation in the field, a hook here) must not leave
    the batch believing it is up to date: round 1 committed the version stamps first, and the next call would have
    launched on stale descriptors.  The next call rebuilds and evaluates the NEW points."""
    P, pr = _problem(hip)
    B = hip.Batch([P])
    g0 = B.eval(Q0, T0)
    P.set_points(pr["xyz"][::2])                    # the batch is dirty now
    B.set_tuning("test_fail_build", 1)
    with pytest.raises(hip.EAError) as ei:
        B.eval(Q0, T0)
    assert ei.value.code == hip.EA_ERR_ALLOC
    g1 = B.eval(Q0, T0)                             # rebuilds: half the points, not the stale descriptors
    Pref = hip.Problem(*pr["K"]); Pref.set_points(pr["xyz"][::2]); Pref.set_dt_grid(pr["grid"])
    want = Pref.eval(Q0, T0)
    assert g1["cost"][0] == want["cost"] and np.array_equal(g1["JtJ"][0], want["JtJ"]) and g1["cost"][0] != g0["cost"][0]
    B.close(); P.close(); Pref.close()


def test_resource_cache_hands_blocks_to_the_next_problem_without_changing_results(hip):
    """ea_problem_destroy / ea_batch_destroy keep their device blocks, pinned blocks and streams for the next owner (what
    makes a problem per ceres::Solve affordable).  A problem built on recycled blocks -- larger, smaller, another dtype,
    after a solve whose queued-ahead launches were still draining -- evaluates and solves to the same bits as the first time;
    ea_release_cached_memory() gives everything back and the next problem still works."""
    import time
    L = hip.load()
    pr_a = synth.make_problem(120, 160, 9000, 40, 1, 130.0, 130.0, 79.5, 59.5, planted_q=synth.quat_from_axis_angle([1, 2, 3], 0.01),
                              planted_t=(0.01, -0.005, 0.02), normalize=True)
    pr_b = synth.make_problem(96, 128, 2500, 12, 5, 120.0, 121.0, 63.5, 47.5, planted_q=synth.quat_from_axis_angle([3, 1, 2], 0.008),
                              planted_t=(0.0, 0.01, 0.005), normalize=True)
    q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)

    def run(pr, dtype):
        P = hip.Problem(*pr["K"], dtype=dtype)
        P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
        g = P.eval(q0, t0)
        q, t, s = P.solve(q0, t0)
        r, J, bad = P.eval_rows(q, t)
        P.close()   # (launches queued past the end of the solve may still be draining here)
        return g["cost"], g["JtJ"].copy(), q.copy(), t.copy(), s["num_iterations"], r.copy()

    assert L.ea_release_cached_memory() == 0
    first = {(k, d): run(pr, d) for k, pr in (("a", pr_a), ("b", pr_b)) for d in (hip.EA_F64, hip.EA_F32)}
    t_fresh = time.perf_counter(); run(pr_a, hip.EA_F64); t_fresh = time.perf_counter() - t_fresh
    order = [("a", hip.EA_F64), ("b", hip.EA_F32), ("b", hip.EA_F64), ("a", hip.EA_F32), ("a", hip.EA_F64), ("b", hip.EA_F64)] * 3
    for k, d in order:
        got = run(pr_a if k == "a" else pr_b, d)
        want = first[(k, d)]
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (k, d)
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]) and got[4] == want[4], (k, d)
        assert np.array_equal(got[5], want[5]), (k, d)
    assert L.ea_release_cached_memory() == 0
    got = run(pr_b, hip.EA_F64)
    assert got[0] == first[("b", hip.EA_F64)][0] and np.array_equal(got[2], first[("b", hip.EA_F64)][2])


def test_recycled_blocks_are_drained_before_their_next_owner(hip):
    """a finished solve leaves (evaluate, step) pairs queued past its end; destroying the problem at once and
    building the next one on the same recycled device blocks must not let those launches meet the new owner's data.  The
    cache drains the device once before the first reuse that follows a free.  Two same-sized problems with different data
    alternate 60 times with six pairs kept queued ahead: every solve lands on the bits of its first run."""
    prs = [synth.make_problem(120, 160, 6000, 40, seed, 130.0, 130.0, 79.5, 59.5, planted_q=synth.quat_from_axis_angle(ax, 0.012),
                              planted_t=tt, normalize=True)
           for seed, ax, tt in ((1, [1, 2, 3], (0.01, -0.005, 0.02)), (2, [3, -1, 2], (-0.01, 0.008, 0.01)))]
    q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)

    def run(pr):
        P = hip.Problem(*pr["K"], dtype=hip.EA_F64)
        P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
        q, t, s = P.solve(q0, t0, iterations_per_sync=6)
        P.close()   # at once: launches queued past the end of the solve are still in the stream
        return q, t, s["num_iterations"], s["final_cost"]

    first = [run(pr) for pr in prs]
    for k in range(60):
        got, want = run(prs[k & 1]), first[k & 1]
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:] == want[2:], k

"""The multi-GPU entry points below Python (include/ea_hip.h, ea_comm_*: RCCL called from librccl directly), as far as a
one-GPU box allows: a ONE-RANK communicator.  What is checked is every call the N-rank run makes -- communicator
creation from a unique id and through ncclCommInitAll, the pose all-gather on the batch's stream, the point-sharded
solve with ncclAllReduce enqueued by the library on the solve's stream -- and their results; two ranks cannot share one
device under RCCL, so the two-rank protocol is covered with gloo (tests/test_gpu_sharded.py, tests/test_dist_gloo.py) and
the xGMI hop itself only by the driver's 8-GPU node.

The C++ program examples/node_batch_demo.cpp is the north_star sentence in the reference's language: per-device batches
solved by host threads, one RCCL gather of the poses."""
import os
import struct
import subprocess

import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q0, T0 = np.array([1.0, 0, 0, 0]), np.zeros(3)


def test_one_hip_runtime_when_torch_comes_first(hip):
    """PyTorch-ROCm ships a libamdhip64 under the same SONAME: imported first (tests/conftest.py), this library binds to
    that copy and the process holds ONE runtime -- streams and device pointers can cross between torch and the library"""
    assert hip.runtime_copies() == 1
    mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
    assert len(mapped) == 1, mapped


def test_pose_gather_and_sharded_solve_on_a_one_rank_communicator(hip):
    cfg = synth.config_c2_twin(seed=17, n_points=30011)
    comm = hip.Comm(hip.comm_unique_id(), 1, 0, device=0)
    try:
        assert comm.info("device") == 0 and hip.load().ea_comm_size(comm._h) == 1 and hip.load().ea_comm_rank(comm._h) == 0
        for dtype, tol in ((hip.EA_F64, 1e-10), (hip.EA_F32, 1e-6)):
            P = hip.Problem(*cfg["K"], dtype=dtype)
            P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
            q, t, s = P.solve(Q0, T0)
            for rows_form in (1, 0):
                # first choice: one launch per iteration, the partial rows all-reduced; EA_SHARDED_ROWS=0: evaluation, fold,
                # all-reduce of the 32 sums, step
                os.environ["EA_SHARDED_ROWS"] = str(rows_form)
                for ahead in (0, 1, 3):
                    before, rows_before = comm.info("allreduces"), comm.info("row_solves")
                    q2, t2, s2 = P.solve_sharded_comm(Q0, T0, comm, iterations_per_sync=ahead)
                    assert comm.info("row_solves") - rows_before == rows_form
                    assert s2["num_iterations"] == s["num_iterations"] and s2["why"] == s["why"], ahead
                    if rows_form:   # on one rank: the very launches of ea_solve
                        assert np.array_equal(q, q2) and np.array_equal(t, t2) and np.array_equal(s2["it_cost"], s["it_cost"])
                    assert np.abs(q - q2).max() < tol and np.abs(t - t2).max() < tol
                    assert s2["it_cost"] == pytest.approx(s["it_cost"], rel=1e-9 if dtype == hip.EA_F64 else 1e-5)
                    # the look-ahead rule: (iteration the solve finished at) + ahead collectives, on every rank alike (the rows
                    # form exchanges once more, behind the evaluation at the start pose)
                    assert comm.info("allreduces") - before == s["num_iterations"] + (ahead or 2) + rows_form, (ahead, rows_form)
            os.environ.pop("EA_SHARDED_ROWS")
            q3, t3, s3 = P.solve_sharded_comm(Q0, T0, comm, strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
            q4, t4, s4 = P.solve(Q0, T0, strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
            assert s3["num_iterations"] == s4["num_iterations"] and np.abs(q3 - q4).max() < tol
            # a solve cut short by the iteration cap ends on every rank alike
            q5, t5, s5 = P.solve_sharded_comm(Q0, T0, comm, max_num_iterations=3)
            q6, t6, s6 = P.solve(Q0, T0, max_num_iterations=3)
            assert s5["num_iterations"] == s6["num_iterations"] == 3 and s5["why"] == s6["why"] and np.abs(q5 - q6).max() < tol
            P.close()
        # the pose gather: behind a batch's solve (its stream) and stand-alone (the communicator's stream)
        Ps = []
        for i in range(3):
            P = hip.Problem(*cfg["K"], dtype=hip.EA_F64)
            P.set_points(cfg["xyz"][i::3]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
            Ps.append(P)
        B = hip.Batch(Ps)
        qs, ts, ss = B.solve(np.tile(Q0, (3, 1)), np.zeros((3, 3)))
        st = [x["termination"] for x in ss]
        for after in (B, None):
            qa, ta, sa = comm.gather_poses(qs, ts, st, after=after)
            assert np.array_equal(qa, qs) and np.array_equal(ta, ts) and list(sa) == st
        qa, ta, sa = comm.gather_poses(qs[:1], ts[:1])            # smaller count re-uses the buffers, status defaults to 0
        assert np.array_equal(qa, qs[:1]) and list(sa) == [0]
        assert comm.info("allgathers") == 3
        B.close()
        for P in Ps:
            P.close()
    finally:
        comm.close()


def test_comm_argument_checks(hip):
    import ctypes as C
    L = hip.load()
    h = C.c_void_p()
    assert L.ea_comm_create(C.byref(h), None, 1, 0, 0) == hip.EA_ERR_INVALID_ARG
    uid = hip.comm_unique_id()
    assert L.ea_comm_create(C.byref(h), uid, 2, 2, 0) == hip.EA_ERR_INVALID_ARG      # rank out of range
    assert L.ea_comm_create(C.byref(h), uid, 1, 0, 99) == hip.EA_ERR_INVALID_ARG     # no such device
    assert L.ea_comm_create_all(C.byref(h), None, 0) == hip.EA_ERR_INVALID_ARG
    devs = (C.c_int * 2)(0, 0)
    two = (C.c_void_p * 2)()
    assert L.ea_comm_create_all(two, devs, 2) == hip.EA_ERR_INVALID_ARG              # a device twice (or no second device)
    assert L.ea_comm_gather_poses(None, None, None, None, None, 1, None, None, None) == hip.EA_ERR_INVALID_ARG
    assert L.ea_solve_sharded_comm(None, None, None, None, None, None) == hip.EA_ERR_INVALID_ARG


def _write_problem(path, aX, grid, K):
    W, H = grid.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", aX.shape[1], H, W))
        f.write(struct.pack("<dddd", *K))
        f.write(np.ascontiguousarray(aX.T, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(grid, dtype=np.float64).tobytes())


def test_cpp_host_threads_solve_per_device_batches_and_gather(hip, bundled_pair, tmp_path):
    """examples/node_batch_demo.cpp on the bundled pair: ea_comm_create_all over the visible devices, one host thread per
    device building and solving its batch, ONE ncclAllGather of the poses; the gathered poses are those the same batch
    solves to through the ctypes path, in rank-major order."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples"), "node_batch_demo"])
    p = str(tmp_path / "pair13.bin")
    aX = bundled_pair["aX"][:, ::7]
    _write_problem(p, aX, bundled_pair["grids"][3], bundled_pair["K"])
    m = 4
    out = subprocess.run([os.path.join(ROOT, "examples", "node_batch_demo"), p, str(m), "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RCCL version" not in out.stdout, out.stdout[:600]   # the library keeps RCCL's banner off the caller's stdout
    lines = out.stdout.strip().splitlines()
    ndev, pairs, solve_ms, gather_ms, same = lines[0].split()
    assert (int(ndev), int(pairs), int(same)) == (1, m, 1) and float(solve_ms) > 0 and float(gather_ms) > 0
    got = np.array([[float(x) for x in l.split()] for l in lines[1:]])
    assert got.shape == (m, 9) and list(got[:, 0]) == list(range(m))
    Ps = []
    for i in range(m):
        P = hip.Problem(*bundled_pair["K"], dtype=hip.EA_F64)
        P.set_points(np.ascontiguousarray(aX.T[:, :3])); P.set_dt_grid(bundled_pair["grids"][3]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    B = hip.Batch(Ps)
    q0 = np.array([[np.cos(0.5 * np.deg2rad(0.02 * g)), 0, 0, np.sin(0.5 * np.deg2rad(0.02 * g))] for g in range(m)])
    qs, ts, ss = B.solve(q0, np.zeros((m, 3)))
    assert np.abs(got[:, 1:5] - qs).max() < 1e-12 and np.abs(got[:, 5:8] - ts).max() < 1e-12
    assert list(got[:, 8].astype(int)) == [x["termination"] for x in ss]
    assert len({tuple(r) for r in got[:, 1:5]}) == m          # the start poses differ, so do the converged bits
    B.close()
    for P in Ps:
        P.close()

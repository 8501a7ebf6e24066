"""The N>1 path on CPU: world_size-2 `gloo` run of the sharding + single pose gather that
bench.py performs over RCCL.  Local "solves" use the oracle (CPU) on tiny planted problems."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    import torch.distributed as dist
    from edge_alignment_amd import dist as ead, synth
    from oracle import ea_oracle as eo
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = ead.shard_indices(n_total, rank, world)
    q_l, t_l, s_l = [], [], []
    for i in mine:
        pr = synth.make_problem(60, 80, 300, 12, 100 + i, 65.0, 65.0, 39.5, 29.5,
                                planted_q=synth.quat_from_axis_angle([1, i + 1, 2], np.deg2rad(0.5)),
                                planted_t=(0.002 * i, -0.001, 0.003), normalize=True)
        P = eo.OracleProblem(pr["grid"], *pr["K"])
        q, t, s = P.solve(pr["xyz"], [1, 0, 0, 0], [0, 0, 0])
        q_l.append(q); t_l.append(t); s_l.append(s["termination"])
    q, t, st = ead.gather_poses(q_l, t_l, s_l, n_total, rank, world)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), q=q, t=t, st=st)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 4])
def test_shard_and_gather_world2(tmp_path, n_total):
    import torch.multiprocessing as mp
    from edge_alignment_amd import dist as ead, synth
    world = 2
    assert sorted(ead.shard_indices(n_total, 0, world) + ead.shard_indices(n_total, 1, world)) == list(range(n_total))
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["q"], r1["q"]) and np.array_equal(r0["t"], r1["t"])
    for i in range(n_total):
        q_true = synth.quat_from_axis_angle([1, i + 1, 2], np.deg2rad(0.5))
        assert synth.rotation_angle_between(r0["q"][i], q_true) < 1e-5
        assert np.linalg.norm(r0["t"][i] - np.array([0.002 * i, -0.001, 0.003])) < 1e-5
        assert r0["st"][i] == 0


def test_single_rank_gather_needs_no_process_group():
    from edge_alignment_amd import dist as ead
    q, t, st = ead.gather_poses([[1, 0, 0, 0], [0, 1, 0, 0]], [[1, 2, 3], [4, 5, 6]], [0, 1], 2, 0, 1)
    assert q.tolist() == [[1, 0, 0, 0], [0, 1, 0, 0]] and t.tolist() == [[1, 2, 3], [4, 5, 6]] and st.tolist() == [0, 1]


# ---- one problem sharded by points: an all-reduce of the 32 accumulator slots per trust-region iteration ---------
# (on the GPU the local evaluation is ea_solve_sharded's fused kernel, here the oracle plays the
# evaluator so the sharding + collective + replicated state machine run on CPU)

def _sharded_worker(rank, world, port, shim_path, out_dir):
    import ctypes as C
    import torch.distributed as dist
    from edge_alignment_amd import dist as ead, synth
    from oracle import ea_oracle as eo
    import test_lm_host_logic as tl
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pr = synth.make_problem(60, 80, 901, 14, 42, 65.0, 65.0, 39.5, 29.5,
                            planted_q=synth.quat_from_axis_angle([1, -2, 0.5], np.deg2rad(0.6)),
                            planted_t=(0.004, -0.002, 0.003), normalize=True)
    X = pr["xyz"][ead.shard_slice(pr["xyz"].shape[0], rank, world)]
    P = eo.OracleProblem(pr["grid"], *pr["K"])
    allreduce = ead.make_allreduce(world)
    shim = C.CDLL(shim_path)

    def cb(pose, acc, _):
        x = np.array([pose[i] for i in range(7)])
        a = np.zeros(32)
        if X.shape[0]:
            e = P.eval(X, x[:4], x[4:])
            a[:21] = e["JtJ"][np.triu_indices(6)]
            a[21:27] = e["Jtr"]; a[27] = e["cost"]; a[28] = float(e["n_invalid"])
        allreduce(a)                      # the one exchange step of an iteration
        for i in range(32):
            acc[i] = a[i]
    out = tl.ShimOut()
    o = tl._opts()
    q0 = np.array([1.0, 0, 0, 0]); t0 = np.zeros(3)
    shim.ea_lm_host_solve.argtypes = [C.POINTER(tl.LMOptions), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, tl.CB,
                                      C.c_void_p, C.POINTER(tl.ShimOut)]
    rc = shim.ea_lm_host_solve(C.byref(o), q0.ctypes.data_as(C.POINTER(C.c_double)), t0.ctypes.data_as(C.POINTER(C.c_double)),
                               0, tl.CB(cb), None, C.byref(out))
    assert rc == 0
    np.savez(os.path.join(out_dir, "sh%d.npz" % rank), x=np.array(out.x[:]), it=out.iteration, why=out.why)
    dist.barrier()
    dist.destroy_process_group()


def test_point_sharded_solve_world2(tmp_path, lm_host_shim, oracle):
    import torch.multiprocessing as mp
    from edge_alignment_amd import dist as ead, synth
    assert [ead.shard_slice(10, r, 3) for r in range(3)] == [slice(0, 3), slice(3, 6), slice(6, 10)]
    world = 2
    mp.spawn(_sharded_worker, args=(world, _free_port(), lm_host_shim._name, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "sh0.npz"), np.load(tmp_path / "sh1.npz")
    # lockstep: both ranks took the same decisions and hold the same pose, bit for bit
    assert np.array_equal(r0["x"], r1["x"]) and r0["it"] == r1["it"] and r0["why"] == r1["why"]
    # and the sharded solve is the unsharded one up to the order of the sums
    pr = synth.make_problem(60, 80, 901, 14, 42, 65.0, 65.0, 39.5, 29.5,
                            planted_q=synth.quat_from_axis_angle([1, -2, 0.5], np.deg2rad(0.6)),
                            planted_t=(0.004, -0.002, 0.003), normalize=True)
    q, t, s = oracle.OracleProblem(pr["grid"], *pr["K"]).solve(pr["xyz"], [1, 0, 0, 0], [0, 0, 0])
    assert r0["it"] == s["num_iterations"]
    assert np.abs(r0["x"][:4] - q).max() < 1e-10 and np.abs(r0["x"][4:] - t).max() < 1e-10


# ---- BASELINE config C4 as bench.py runs it: a block of frame pairs per rank, one batched solve, ONE all-gather from
# preallocated tensors (edge_alignment_amd.dist.run_c4 / PoseGather).  On CPU the batched solve is the oracle on small
# planted stand-ins keyed by the C4 specification (frame pair, start pose); the sharding, the start poses, the gather
# and the global order are the code bench.py runs over RCCL.

def _c4_standin(spec):
    from edge_alignment_amd import synth
    a, b, q0, t0 = spec
    return synth.make_problem(60, 80, 250, 12, 1000 + 10 * a + b, 65.0, 65.0, 39.5, 29.5,
                              planted_q=synth.quat_from_axis_angle([a, b, 1], np.deg2rad(0.4)),
                              planted_t=(0.001 * a, -0.001 * b, 0.002), normalize=True)


def _c4_worker(rank, world, port, per_gpu, out_dir):
    import torch.distributed as dist
    from edge_alignment_amd import dist as ead
    from oracle import ea_oracle as eo
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)

    def build_and_solve(specs):
        probs = [_c4_standin(sp) for sp in specs]

        def solve():
            qs, ts, ss = [], [], []
            for pr, sp in zip(probs, specs):
                q, t, s = eo.OracleProblem(pr["grid"], *pr["K"]).solve(pr["xyz"], sp[2], sp[3])
                qs.append(q); ts.append(t); ss.append(s)
            return np.array(qs), np.array(ts), ss
        return solve, {"dtype": "f64"}
    res, (q, t, st) = ead.run_c4(rank, world, build_and_solve, per_gpu=per_gpu, device="cpu")
    assert res["pairs_total"] == per_gpu * world and res["converged"] == per_gpu
    np.savez(os.path.join(out_dir, "c4_rank%d.npz" % rank), q=q, t=t, st=st)
    dist.barrier()
    dist.destroy_process_group()


def test_c4_run_shape_world2(tmp_path):
    import torch.multiprocessing as mp
    from edge_alignment_amd import dist as ead, synth
    world, per_gpu = 2, 3
    assert ead.shard_block(6, 0, 2) == [0, 1, 2] and ead.shard_block(6, 1, 2) == [3, 4, 5]
    specs = synth.config_c4_specs(total=per_gpu * world)
    assert specs == [] or (np.array_equal(specs[0][2], [1, 0, 0, 0]) and not np.any(specs[0][3]))  # identity start first
    full = synth.config_c4_specs()
    assert len(full) == 256 and len({(a, b) for a, b, _, _ in full}) == 20
    ang = [synth.rotation_angle_between(q, [1, 0, 0, 0]) for _, _, q, _ in full]
    assert max(ang) <= np.deg2rad(1.0) + 1e-12 and max(np.linalg.norm(t) for _, _, _, t in full) <= 0.02 + 1e-12
    mp.spawn(_c4_worker, args=(world, _free_port(), per_gpu, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "c4_rank0.npz"), np.load(tmp_path / "c4_rank1.npz")
    assert np.array_equal(r0["q"], r1["q"]) and np.array_equal(r0["t"], r1["t"]) and np.array_equal(r0["st"], r1["st"])
    assert r0["q"].shape == (6, 4)
    for i, sp in enumerate(specs):  # global order = specification order
        a, b = sp[0], sp[1]
        assert synth.rotation_angle_between(r0["q"][i], synth.quat_from_axis_angle([a, b, 1], np.deg2rad(0.4))) < 1e-5
        assert np.linalg.norm(r0["t"][i] - np.array([0.001 * a, -0.001 * b, 0.002])) < 1e-5
        assert r0["st"][i] == 0


def test_pose_gather_single_rank_is_a_copy():
    from edge_alignment_amd import dist as ead
    pg = ead.PoseGather(2, 1)
    q, t, st = pg.gather([[1, 0, 0, 0], [0, 1, 0, 0]], [[1, 2, 3], [4, 5, 6]], [0, 2])
    assert q.tolist() == [[1, 0, 0, 0], [0, 1, 0, 0]] and t.tolist() == [[1, 2, 3], [4, 5, 6]] and st.tolist() == [0, 2]


def _barrier_worker(rank, world, port, out_dir):
    import time
    import torch.distributed as dist
    from edge_alignment_amd import dist as ead
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    nb = ead.NodeBarrier(rank, world)
    # a token only the last rank to arrive can have written must be visible to everybody who leaves the barrier
    from multiprocessing import shared_memory  # noqa: F401  (the barrier's own segment carries the token: slot column 1)
    late = 0
    rng = np.random.default_rng(rank)
    t0 = time.perf_counter()
    rounds = 3000
    for k in range(1, rounds + 1):
        if rng.random() < 0.02:
            time.sleep(0.0005)
        nb._slots[rank, 1] = k          # written before entering barrier k
        nb.wait()
        if not (nb._slots[:, 1] >= k).all():
            late += 1
    el = time.perf_counter() - t0
    np.savez(os.path.join(out_dir, "bar%d.npz" % rank), late=late, us=el / rounds * 1e6, epoch=nb.epoch)
    dist.barrier()
    nb.close()
    dist.destroy_process_group()


def test_node_barrier_world3(tmp_path):
    """the shared-memory barrier of bench.py's timed bracket: nobody leaves barrier k before everybody has entered it"""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_barrier_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = np.load(tmp_path / ("bar%d.npz" % r))
        assert int(d["late"]) == 0 and int(d["epoch"]) == 3000
        assert float(d["us"]) < 2000.0

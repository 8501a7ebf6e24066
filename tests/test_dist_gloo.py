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

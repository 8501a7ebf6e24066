"""One problem sharded by points: ea_solve_sharded = local fused evaluation on the device, an
all-reduce of the 32 accumulator slots per iteration, the trust-region step replicated on every rank."""
import os
import socket

import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
Q0, T0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
QFAR, TFAR = np.array([np.cos(0.1), 0.6 * np.sin(0.1), 0.0, 0.8 * np.sin(0.1)]), np.array([0.05, -0.04, 0.03])  # 11.5 degrees off


def _problem():
    return synth.config_c2_twin(seed=17, n_points=30011)


def test_sharded_solve_single_rank_equals_device_solve(hip):
    """world size 1: the all-reduce is the identity; the host-side replica of the state machine must take the same
    decisions as the device LM-step kernel."""
    from edge_alignment_amd import dist as ead
    cfg = _problem()
    for dtype, tol in ((hip.EA_F64, 1e-10), (hip.EA_F32, 1e-6)):
        P = hip.Problem(*cfg["K"], dtype=dtype)
        P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
        q, t, s = P.solve(Q0, T0)
        calls = []
        ident = ead.make_allreduce(1)
        def ar(a):
            calls.append(a.shape[0]); ident(a)
        q2, t2, s2 = P.solve_sharded(Q0, T0, ar)
        assert s2["num_iterations"] == s["num_iterations"] and s2["why"] == s["why"]
        assert np.abs(q - q2).max() < tol and np.abs(t - t2).max() < tol
        assert calls == [32] * (s["num_iterations"] + 1)   # one exchange per evaluation
        q3, t3, s3 = P.solve_sharded(Q0, T0, ar, strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
        q4, t4, s4 = P.solve(Q0, T0, strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
        assert s3["num_iterations"] == s4["num_iterations"] and np.abs(q3 - q4).max() < tol
        P.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from edge_alignment_amd import capi, dist as ead
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg = _problem()
    X = cfg["xyz"][ead.shard_slice(cfg["xyz"].shape[0], rank, world)]
    P = capi.Problem(*cfg["K"], dtype=capi.EA_F64, device=0)   # both ranks share the one GPU of the test box
    P.set_points(X); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    q, t, s = P.solve_sharded(Q0, T0, ead.make_allreduce(world))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), q=q, t=t, it=s["num_iterations"], cost=s["final_cost"])
    P.close()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_solve_two_ranks(hip, tmp_path):
    import torch.multiprocessing as mp
    cfg = _problem()
    P = hip.Problem(*cfg["K"], dtype=hip.EA_F64)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
    q, t, s = P.solve(Q0, T0)
    P.close()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["q"], r1["q"]) and np.array_equal(r0["t"], r1["t"]) and r0["it"] == r1["it"]   # lockstep
    assert r0["it"] == s["num_iterations"]
    assert np.abs(r0["q"] - q).max() < 1e-10 and np.abs(r0["t"] - t).max() < 1e-10
    assert r0["cost"] == pytest.approx(s["final_cost"], rel=1e-10)


# ---- the exchange kept on the stream: ea_solve_sharded_device (evaluation -> fold into the caller's device buffer ->
# the caller's collective enqueued on the library's stream -> the device step kernel reading that buffer)

def test_sharded_device_single_rank_equals_device_solve(hip):
    """world size 1: nothing to enqueue; evaluation, fold, step all stay on the device and must take the decisions of
    ea_solve (the rows are folded by the fold kernel instead of the step kernel: sums equal to rounding)."""
    import torch
    from edge_alignment_amd import dist as ead
    cfg = _problem()
    for dtype, tol in ((hip.EA_F64, 1e-10), (hip.EA_F32, 1e-6)):
        P = hip.Problem(*cfg["K"], dtype=dtype)
        P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
        q, t, s = P.solve(Q0, T0)
        sums, enqueue = ead.make_device_allreduce(1, torch.device("cuda", 0))
        calls = []
        def enq(stream):
            calls.append(stream); enqueue(stream)
        for per_sync in (0, 1, 3):
            del calls[:]
            q2, t2, s2 = P.solve_sharded_device(Q0, T0, enq, sums.data_ptr(), iterations_per_sync=per_sync)
            assert s2["num_iterations"] == s["num_iterations"] and s2["why"] == s["why"], per_sync
            assert np.abs(q - q2).max() < tol and np.abs(t - t2).max() < tol
            assert s2["it_cost"] == pytest.approx(s["it_cost"], rel=1e-9 if dtype == hip.EA_F64 else 1e-5)
            # the look-ahead rule: (evaluations of the solve - 1) + `ahead` iterations enqueued -- a function of where the
            # solve finished, the count every rank reaches alike
            assert len(calls) == s["num_iterations"] + (per_sync or 2)
            assert len(set(calls)) == 1                     # one stream
        q3, t3, s3 = P.solve_sharded_device(Q0, T0, enq, sums.data_ptr(), strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
        q4, t4, s4 = P.solve(Q0, T0, strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25)
        assert s3["num_iterations"] == s4["num_iterations"] and np.abs(q3 - q4).max() < tol
        P.close()


def _device_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    torch.cuda.init()  # before libea_hip touches the device (conftest.py `hip`)
    from edge_alignment_amd import capi, dist as ead
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg = _problem()
    X = cfg["xyz"][ead.shard_slice(cfg["xyz"].shape[0], rank, world)]
    P = capi.Problem(*cfg["K"], dtype=capi.EA_F64, device=0)   # both ranks share the one GPU of the test box
    P.set_points(X); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    sums, enqueue = ead.make_device_allreduce(world, torch.device("cuda", 0))
    n = [0]
    def enq(stream):
        n[0] += 1; enqueue(stream)
    q, t, s = P.solve_sharded_device(Q0, T0, enq, sums.data_ptr(), solve_timeout_ms=20000.0)
    np.savez(os.path.join(out_dir, "d%d.npz" % rank), q=q, t=t, it=s["num_iterations"], cost=s["final_cost"], calls=n[0])
    P.close()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_device_two_ranks(hip, tmp_path):
    """Two gloo ranks on the one GPU: the collective is staged through the host here (gloo has no device path), the
    protocol is the RCCL one -- the look-ahead rule, the same number of collectives on every rank, lockstep iterates."""
    import torch.multiprocessing as mp
    cfg = _problem()
    P = hip.Problem(*cfg["K"], dtype=hip.EA_F64)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
    q, t, s = P.solve(Q0, T0)
    P.close()
    mp.spawn(_device_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "d0.npz"), np.load(tmp_path / "d1.npz")
    assert np.array_equal(r0["q"], r1["q"]) and np.array_equal(r0["t"], r1["t"]) and r0["it"] == r1["it"]   # lockstep
    assert r0["calls"] == r1["calls"]                                                                       # collectives match
    assert r0["it"] == s["num_iterations"]
    assert np.abs(r0["q"] - q).max() < 1e-10 and np.abs(r0["t"] - t).max() < 1e-10
    assert r0["cost"] == pytest.approx(s["final_cost"], rel=1e-10)


class _DeviceDoubles:
    """`count` doubles of device memory at `ptr` as an object torch can wrap (the library owns the rows it has all-reduced)"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _rows_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    from edge_alignment_amd import capi
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg = _problem()
    n = cfg["xyz"].shape[0]
    cut = [0, 20000, n] if world == 2 else [0, n]          # uneven shards: 79 and 40 partial rows
    X = cfg["xyz"][cut[rank]:cut[rank + 1]]
    P = capi.Problem(*cfg["K"], dtype=capi.EA_F64, device=0)   # both ranks share the one GPU of the test box
    P.set_points(X); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    calls, counts = [0], set()

    def enqueue(ptr, count, stream):
        # gloo has no device path: the rows are staged through the host under a synchronisation of the library's stream
        calls[0] += 1; counts.add(count)
        ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda", 0))
        with torch.cuda.stream(ext):
            rows = torch.as_tensor(_DeviceDoubles(ptr, count), device=torch.device("cuda", 0))
            h = rows.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            rows.copy_(h)

    def agree(vals):
        tt = torch.tensor(vals, dtype=torch.int32)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return [int(tt[0]), int(tt[1])]
    out = {}
    for name, opts in (("lm", {}), ("dogleg", dict(strategy=capi.STRATEGY_DOGLEG, max_num_iterations=25)), ("capped", dict(max_num_iterations=3)),
                       ("picky", dict(min_relative_decrease=0.97, max_num_iterations=12, initial_trust_region_radius=1e8, far=1))):
        calls[0] = 0
        far = opts.pop("far", 0)
        q, t, s, used = P.solve_sharded_rows(QFAR if far else Q0, TFAR if far else T0, enqueue, agree, solve_timeout_ms=20000.0, **opts)
        out.update({name + "_q": q, name + "_t": t, name + "_it": s["num_iterations"], name + "_cost": s["final_cost"],
                    name + "_calls": calls[0], name + "_used": used, name + "_rej": s["num_unsuccessful_steps"]})
    out["counts"] = sorted(counts)
    np.savez(os.path.join(out_dir, "w%d.npz" % rank), **out)
    P.close()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rows_exchange_two_ranks(hip, tmp_path):
    """The one-launch-per-iteration form of the sharded solve (what ea_solve_sharded_comm runs over RCCL), rehearsed over gloo
    with two ranks on the one GPU and UNEVEN shards (79 and 40 partial rows: both ranks fold 79, the shorter one keeps zeros
    behind its own): lockstep iterates, the same number of exchanges on both ranks, the whole problem's solution."""
    import torch.multiprocessing as mp
    cfg = _problem()
    P = hip.Problem(*cfg["K"], dtype=hip.EA_F64)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(hip.LOSS_CAUCHY, 1.0)
    ref = {"lm": P.solve(Q0, T0), "dogleg": P.solve(Q0, T0, strategy=hip.STRATEGY_DOGLEG, max_num_iterations=25),
           "capped": P.solve(Q0, T0, max_num_iterations=3),
           "picky": P.solve(QFAR, TFAR, min_relative_decrease=0.97, max_num_iterations=12, initial_trust_region_radius=1e8)}
    P.close()
    mp.spawn(_rows_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "w0.npz"), np.load(tmp_path / "w1.npz")
    assert list(r0["counts"]) == list(r1["counts"]) == [79 * 32]
    for name, (q, t, s) in ref.items():
        assert r0[name + "_used"] == 1 and r1[name + "_used"] == 1, name
        assert np.array_equal(r0[name + "_q"], r1[name + "_q"]) and np.array_equal(r0[name + "_t"], r1[name + "_t"]), name   # lockstep
        assert r0[name + "_it"] == r1[name + "_it"] == s["num_iterations"], name
        assert r0[name + "_calls"] == r1[name + "_calls"] == s["num_iterations"] + 2 + 1, name   # prologue + look-ahead rule
        assert np.abs(r0[name + "_q"] - q).max() < 1e-10 and np.abs(r0[name + "_t"] - t).max() < 1e-10, name
        assert r0[name + "_cost"] == pytest.approx(s["final_cost"], rel=1e-10), name
        assert r0[name + "_rej"] == r1[name + "_rej"] == s["num_unsuccessful_steps"], name
    # (rejected steps -- the cold system travelling through the launches -- are exercised by tests/test_gpu_fused_iterations.py)


def _rccl_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    torch.cuda.set_device(0)
    from edge_alignment_amd import capi, dist as ead
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    cfg = _problem()
    P = capi.Problem(*cfg["K"], dtype=capi.EA_F64, device=0)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    q, t, s = P.solve(Q0, T0)
    sums, enqueue = ead.make_device_allreduce(world, torch.device("cuda", 0), force_collective=True)
    n = [0]
    def enq(stream):
        n[0] += 1; enqueue(stream)
    q2, t2, s2 = P.solve_sharded_device(Q0, T0, enq, sums.data_ptr(), solve_timeout_ms=20000.0)
    g = ead.PoseGather(3, world, device=torch.device("cuda", 0), force_collective=True)
    qs = np.stack([q2, q, q2]); ts = np.stack([t2, t, t2])
    gq, gt, gs = g.gather(qs, ts, [0, 1, 2])
    dist.barrier()
    np.savez(os.path.join(out_dir, "rccl.npz"), q=q, t=t, q2=q2, t2=t2, it=s["num_iterations"], it2=s2["num_iterations"], calls=n[0],
             gq=gq, gt=gt, gs=gs, qs=qs, ts=ts, backend=dist.get_backend())
    P.close()
    dist.destroy_process_group()


def test_rccl_calls_on_a_one_rank_group(hip, tmp_path):
    """The collectives of both multi-GPU modes through RCCL itself (backend "nccl"), as far as a one-GPU box allows: a
    one-rank group, the all-reduce enqueued on the library's stream under torch's ExternalStream inside
    ea_solve_sharded_device, and the pose all_gather_into_tensor from the preallocated device tensors.  (Two ranks cannot
    share one device under RCCL; the two-rank protocol is the gloo test above.)"""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert str(r["backend"]) == "nccl"
    assert r["it2"] == r["it"] and r["calls"] >= r["it"] + 1
    assert np.abs(r["q2"] - r["q"]).max() < 1e-10 and np.abs(r["t2"] - r["t"]).max() < 1e-10
    assert np.array_equal(r["gq"], r["qs"]) and np.array_equal(r["gt"], r["ts"]) and list(r["gs"]) == [0, 1, 2]

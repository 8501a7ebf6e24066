"""Multi-GPU layer of the Python harness: one process per GPU, independent frame-pair problems sharded in blocks,
ONE collective — an all-gather of the solved poses (RCCL over xGMI).  A single large problem can instead be sharded by
points (`shard_slice`): one all-reduce of 32 doubles per trust-region iteration, the step itself replicated on every rank.

The collectives themselves live BELOW Python, in the C-ABI (include/ea_hip.h, ea_comm_*: ncclAllGather / ncclAllReduce
called from librccl on the library's stream): `capi.Comm.gather_poses` and `Problem.solve_sharded_comm` are what a GPU
run uses, and what a C++ caller of the library uses without any of this file (examples/node_batch_demo.cpp).  What is
left here: the sharding arithmetic, the run shape of BASELINE config C4, the shared-memory barrier of bench.py's timed
bracket, and -- for rehearsals on one GPU or on the CPU, where RCCL cannot hold two ranks -- the torch.distributed forms
of the same exchanges (`make_allreduce`: host-staged, any backend; `make_device_allreduce`: enqueued on the library's
stream through torch's ExternalStream).

The reference is single-process and has no communication at all; frame pairs are
independent units, so no data-path collective exists.  The gather moves 8 doubles per problem
(q wxyz, t, termination) — latency-bound, so it is issued once, after all local solves.
"""
import numpy as np


def shard_slice(n_points, rank, world_size):
    """contiguous point shard of ONE problem: rank r owns [r n / W, (r+1) n / W)  """
    return slice((rank * n_points) // world_size, ((rank + 1) * n_points) // world_size)


def make_allreduce(world_size, device="cpu", force_collective=False):
    """In-place sum of a small float64 numpy array over all ranks: the per-iteration exchange of a point-sharded solve
    (32 accumulator slots = 256 bytes).  RCCL when the process group's backend is "nccl" (device = this rank's GPU),
    gloo on the CPU.  world_size 1: identity, no process group needed."""
    if world_size == 1 and not force_collective:
        return lambda a: None
    import torch
    import torch.distributed as dist
    stage = torch.zeros(64, dtype=torch.float64, device=device)

    def allreduce(a):
        n = a.shape[0]
        stage[:n].copy_(torch.from_numpy(a))
        dist.all_reduce(stage[:n], op=dist.ReduceOp.SUM)
        a[:] = stage[:n].cpu().numpy()
    return allreduce


def make_device_allreduce(world_size, device, force_collective=False):
    """The on-stream exchange of Problem.solve_sharded_device: returns (sums, enqueue) where `sums` is a torch tensor of
    32 doubles on `device` (the buffer the library folds into and the step kernel reads) and `enqueue(stream_ptr)`
    enqueues the in-place all-reduce of it on the library's HIP stream -- torch.distributed under an ExternalStream:
    ProcessGroupNCCL orders its RCCL call behind the work already on that stream and makes the stream wait for the
    result, all on the device, no host synchronisation.  world_size 1: nothing to enqueue.  With a CPU backend (gloo)
    the tensor is staged through the host under a stream synchronisation (the protocol, not the speed, is what a
    gloo rehearsal on one GPU checks)."""
    import torch
    import torch.distributed as dist
    from . import capi
    # The library's stream handle is handed to torch below.  That is only meaningful when both sit on ONE HIP runtime:
    # PyTorch-ROCm ships its own libamdhip64 under the SONAME this library links, so importing torch first gives one copy
    # (checked here); the other load order maps two, and a stream of one is not an object of the other.
    if capi.runtime_copies() != 1:
        raise RuntimeError("%d HIP runtimes are mapped in this process: import torch before edge_alignment_amd.capi loads libea_hip.so "
                           "(or use Problem.solve_sharded, whose exchange is staged through the host)" % capi.runtime_copies())
    sums = torch.zeros(32, dtype=torch.float64, device=device)
    if world_size == 1 and not force_collective:   # (force_collective: rehearse the RCCL call on a one-rank group)
        return sums, (lambda stream_ptr: None)
    backend = dist.get_backend()
    streams = {}

    def enqueue(stream_ptr):
        ext = streams.get(stream_ptr)
        if ext is None:
            ext = streams[stream_ptr] = torch.cuda.ExternalStream(int(stream_ptr), device=sums.device)
        with torch.cuda.stream(ext):
            if backend == "nccl":
                dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            else:
                h = sums.cpu()  # orders behind the fold on `ext`
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                sums.copy_(h)
    return sums, enqueue


def shard_indices(n_items, rank, world_size):
    """problem i -> rank (i mod world_size)"""
    return list(range(rank, n_items, world_size))


def gather_poses(q_local, t_local, status_local, n_total, rank, world_size, device="cpu"):
    """All-gather the locally solved poses.  Returns (q (n_total,4), t (n_total,3), status (n_total,))
    in global problem order on every rank."""
    import torch
    import torch.distributed as dist
    m_max = (n_total + world_size - 1) // world_size
    buf = torch.zeros((m_max, 8), dtype=torch.float64)
    m = len(q_local)
    if m:
        buf[:m, 0:4] = torch.as_tensor(np.asarray(q_local, dtype=np.float64).reshape(m, 4))
        buf[:m, 4:7] = torch.as_tensor(np.asarray(t_local, dtype=np.float64).reshape(m, 3))
        buf[:m, 7] = torch.as_tensor(np.asarray(status_local, dtype=np.float64).reshape(m))
    buf = buf.to(device)
    out = torch.empty((world_size, m_max, 8), dtype=torch.float64, device=device)
    if world_size > 1:
        dist.all_gather_into_tensor(out.view(-1, 8), buf)
    else:
        out[0] = buf
    out = out.cpu().numpy()
    q = np.zeros((n_total, 4)); t = np.zeros((n_total, 3)); st = np.zeros(n_total)
    for r in range(world_size):
        idx = shard_indices(n_total, r, world_size)
        q[idx] = out[r, :len(idx), 0:4]
        t[idx] = out[r, :len(idx), 4:7]
        st[idx] = out[r, :len(idx), 7]
    return q, t, st


def shard_block(n_items, rank, world_size):
    """contiguous block of problems per rank (BASELINE C4: 256 frame pairs, 32 per GPU): rank r owns
    [r n / W, (r+1) n / W)"""
    return list(range((rank * n_items) // world_size, ((rank + 1) * n_items) // world_size))


class PoseGather:
    """The one collective of the batch mode, as a reusable object: a pinned host staging block and two device tensors
    allocated ONCE; gather() is one host-to-device copy of m x 8 doubles, ONE all_gather_into_tensor (RCCL over xGMI
    when the backend is "nccl") and one copy back.  Blocks of `m` problems per rank, global order = rank-major."""

    def __init__(self, m, world_size, device="cpu", force_collective=False, comm=None):
        import torch
        self.m, self.world, self.device = int(m), int(world_size), device
        self.comm = comm   # capi.Comm: the gather is then ea_comm_gather_poses (ncclAllGather issued by the library)
        self.collective = self.world > 1 or force_collective
        self.host = torch.zeros((self.m, 8), dtype=torch.float64)
        if str(device) != "cpu":
            self.host = self.host.pin_memory()
        self.send = torch.zeros((self.m, 8), dtype=torch.float64, device=device)
        self.recv = torch.zeros((self.world * self.m, 8), dtype=torch.float64, device=device)

    def gather(self, q_local, t_local, status_local, after=None):
        """-> (q (W m, 4), t (W m, 3), status (W m,)) on every rank; `after`: the capi.Batch whose solve produced the poses"""
        if self.comm is not None:
            qa, ta, sa = self.comm.gather_poses(q_local, t_local, status_local, after=after)
            return qa, ta, sa.astype(np.float64)
        import torch
        import torch.distributed as dist
        h = self.host.numpy()
        h[:, 0:4] = np.asarray(q_local, dtype=np.float64).reshape(self.m, 4)
        h[:, 4:7] = np.asarray(t_local, dtype=np.float64).reshape(self.m, 3)
        h[:, 7] = np.asarray(status_local, dtype=np.float64).reshape(self.m)
        self.send.copy_(self.host, non_blocking=True)
        if self.collective:
            dist.all_gather_into_tensor(self.recv, self.send)
        else:
            self.recv.copy_(self.send)
        out = self.recv.cpu().numpy()
        return out[:, 0:4].copy(), out[:, 4:7].copy(), out[:, 7].copy()


def run_c4(rank, world_size, build_and_solve, per_gpu=32, device="cpu", total=None, repeats=1, force_collective=False, comm=None):
    """BASELINE config C4 as a run shape: every rank takes its block of `per_gpu` frame-pair problems
    (synth.config_c4_specs), solves them with ONE batched solve (`build_and_solve(specs) -> (solve_fn, info)`,
    solve_fn() -> (q, t, summaries) -- ea_batch_solve on the GPU, a CPU stand-in in the gloo test), then ONE
    all-gather of the per_gpu x 8 doubles (`comm`: through the C-ABI's RCCL communicator; otherwise torch.distributed on
    `device`).  Returns a dict of timings and the gathered poses (global order)."""
    import time
    from . import synth
    total = total if total is not None else per_gpu * world_size
    specs_all = synth.config_c4_specs(total=total)
    mine = shard_block(total, rank, world_size)
    assert len(mine) == per_gpu, "C4 shards evenly: %d problems over %d ranks" % (total, world_size)
    solve_fn, info = build_and_solve([specs_all[i] for i in mine])
    pg = PoseGather(per_gpu, world_size, device=device, force_collective=force_collective, comm=comm)
    q, t, ss = solve_fn()  # warm-up (descriptor upload, first-touch)
    pg.gather(q, t, [s["termination"] for s in ss])
    t0 = time.perf_counter()
    for _ in range(repeats):
        q, t, ss = solve_fn()
    solve_s = (time.perf_counter() - t0) / repeats
    t1 = time.perf_counter()
    qa, ta, sa = pg.gather(q, t, [s["termination"] for s in ss])
    gather_s = time.perf_counter() - t1
    its = sum(s["num_iterations"] for s in ss)
    evals = sum(s.get("num_point_evals", 0) for s in ss)
    out = dict(info)
    out.update({"pose_gather": "ea_comm_gather_poses (ncclAllGather from librccl)" if comm is not None else "torch.distributed all_gather_into_tensor",
                "pairs_per_gpu": per_gpu, "pairs_total": total, "solve_ms": solve_s * 1e3,
                "lm_iters_per_s_per_gpu": its / solve_s, "evals_per_s_per_gpu": evals / solve_s,
                "iterations_mean": its / per_gpu, "pose_gather_ms": gather_s * 1e3,
                "converged": int(sum(1 for s in ss if s["termination"] == 0))})
    return out, (qa, ta, sa)


class NodeBarrier:
    """Barrier for the ranks of ONE node (one process per GPU) through POSIX shared memory: every rank owns a 64-byte line
    holding the number of the last barrier it has entered; a rank leaves barrier k when every line reads >= k.  A few
    microseconds per round instead of the tens a collective-based barrier costs -- which matters where the barrier sits
    INSIDE a timed bracket of ~100 us (bench.py: K = 20 steps of ~3-5 us).  Needs a process group once, for the
    rendezvous (name of the segment, a check that all ranks share a host); raises on anything unexpected so that the
    caller can fall back to torch.distributed.barrier()."""

    def __init__(self, rank, world_size):
        import os
        import secrets
        import socket
        from multiprocessing import resource_tracker, shared_memory
        import torch.distributed as dist
        hosts = [None] * world_size
        dist.all_gather_object(hosts, socket.gethostname())
        if len(set(hosts)) != 1:
            raise RuntimeError("ranks live on different hosts: %s" % sorted(set(hosts)))
        name = ["ea_barrier_%d_%s" % (os.getppid(), secrets.token_hex(4))] if rank == 0 else [None]
        if rank == 0:
            self._shm = shared_memory.SharedMemory(name=name[0], create=True, size=64 * world_size)
            self._shm.buf[:] = bytes(64 * world_size)
        dist.broadcast_object_list(name, src=0)   # (also orders the creation before the attaches)
        if rank != 0:
            self._shm = shared_memory.SharedMemory(name=name[0])
            try:  # the creator unlinks; an attaching process must not (Python's resource tracker would, at its exit)
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self._slots = np.ndarray((world_size, 8), dtype=np.int64, buffer=self._shm.buf)
        self.rank, self.world_size, self.epoch = rank, world_size, 0
        dist.barrier()   # every rank is attached before the first wait

    def wait(self, timeout_s=120.0):
        import time
        self.epoch += 1
        self._slots[self.rank, 0] = self.epoch
        col = self._slots[:, 0]
        t0 = None
        while not (col >= self.epoch).all():
            if t0 is None:
                t0 = time.perf_counter()
            elif time.perf_counter() - t0 > timeout_s:
                raise TimeoutError("NodeBarrier: rank(s) %s missing at barrier %d" % (np.nonzero(col < self.epoch)[0].tolist(), self.epoch))

    def close(self):
        shm, self._shm = getattr(self, "_shm", None), None
        if shm is None:
            return
        self._slots = None
        shm.close()
        if self.rank == 0:
            try:
                shm.unlink()
            except FileNotFoundError:
                pass

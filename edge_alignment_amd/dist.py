"""Multi-GPU layer: one process per GPU, independent frame-pair problems sharded round-robin,
ONE collective — an all-gather of the solved poses (RCCL over xGMI when the backend is "nccl").
A single large problem can instead be sharded by points (`shard_slice`, `make_allreduce`, `Problem.solve_sharded`):
one all-reduce of 32 doubles per trust-region iteration, the step itself replicated on every rank.

The reference is single-process and has no communication at all; frame pairs are
independent units, so no data-path collective exists.  The gather moves 8 doubles per problem
(q wxyz, t, termination) — latency-bound, so it is issued once, after all local solves.
"""
import numpy as np


def shard_slice(n_points, rank, world_size):
    """contiguous point shard of ONE problem: rank r owns [r n / W, (r+1) n / W)  """
    return slice((rank * n_points) // world_size, ((rank + 1) * n_points) // world_size)


def make_allreduce(world_size, device="cpu"):
    """In-place sum of a small float64 numpy array over all ranks: the per-iteration exchange of a point-sharded solve
    (32 accumulator slots = 256 bytes).  RCCL when the process group's backend is "nccl" (device = this rank's GPU),
    gloo on the CPU.  world_size 1: identity, no process group needed."""
    if world_size == 1:
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

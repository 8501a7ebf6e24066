"""Workload for rocprofv3 --kernel-trace: the 1e5-point fp64 solve, 20 solves each of
  ea_solve, one launch per iteration (ea_lm_iter_kernel) and as (evaluate, step) pairs (tuning key "fused_iterations" 0);
  ea_solve_sharded_comm on a one-rank RCCL communicator, rows exchanged (one launch + one all-reduce per iteration) and in the
  sums form (evaluation, fold, ncclAllReduce of 32 doubles, step; EA_SHARDED_ROWS=0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from edge_alignment_amd import capi, synth
cfg = synth.config_c2_twin(seed=7, n_points=100000)
q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
P = capi.Problem(*cfg["K"], dtype=capi.EA_F64)
P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
comm = capi.Comm(capi.comm_unique_id(), 1, 0, device=0)
B = capi.Batch([P])


def pairs():
    B.set_tuning("fused_iterations", 0)
    q, t, s = B.solve(q0, t0)
    return q[0], t[0], s[0]


def sums_form():
    os.environ["EA_SHARDED_ROWS"] = "0"
    try:
        return P.solve_sharded_comm(q0, t0, comm)
    finally:
        os.environ.pop("EA_SHARDED_ROWS")


for name, fn in (("ea_solve (one launch per iteration)", lambda: P.solve(q0, t0)), ("ea_solve as (evaluate, step) pairs", pairs),
                 ("ea_solve_sharded_comm (rows exchanged)", lambda: P.solve_sharded_comm(q0, t0, comm)),
                 ("ea_solve_sharded_comm (sums form)", sums_form)):
    fn()
    ms, its = 0.0, 0
    for _ in range(20):
        q, t, s = fn()
        ms += s["total_time_ms"]; its += s["num_iterations"]
    print("%s: %.1f us per solve, %d iterations, %.3g it/s (library clock)" % (name, ms / 20 * 1e3, its // 20, its / (ms * 1e-3)), flush=True)
B.close(); comm.close(); P.close()

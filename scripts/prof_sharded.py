"""Workload for rocprofv3 --kernel-trace: the 1e5-point fp64 solve through ea_solve (two launches per iteration) and through
ea_solve_sharded_comm on a one-rank RCCL communicator (evaluation, fold, ncclAllReduce, step per iteration), 20 solves each."""
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
for name, fn in (("ea_solve", lambda: P.solve(q0, t0)), ("ea_solve_sharded_comm", lambda: P.solve_sharded_comm(q0, t0, comm))):
    fn()
    ms, its = 0.0, 0
    for _ in range(20):
        q, t, s = fn()
        ms += s["total_time_ms"]; its += s["num_iterations"]
    print("%s: %.1f us per solve, %d iterations, %.3g it/s (library clock)" % (name, ms / 20 * 1e3, its // 20, its / (ms * 1e-3)), flush=True)
comm.close(); P.close()

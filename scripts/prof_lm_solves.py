"""40 device-resident LM solves of the 1e5-point problem (fp64) for a kernel trace."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
cfg = synth.config_c2_twin(seed=7, n_points=100000)
P = capi.Problem(*cfg["K"], dtype=capi.EA_F64); P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
q0, t0 = np.array([1., 0, 0, 0]), np.zeros(3)
for _ in range(40):
    q, t, s = P.solve(q0, t0)
print(s["num_iterations"], s["total_time_ms"])

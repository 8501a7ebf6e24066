import sys, numpy as np
sys.path.insert(0, '.')
from edge_alignment_amd import capi, synth
q0 = np.array([1.,0,0,0]); t0 = np.zeros(3)
cfg = synth.config_c2_twin(seed=7, n_points=100000)
P = capi.Problem(*cfg['K'], dtype=capi.EA_F64); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid'])
for _ in range(5):
    q, t, s = P.solve(q0, t0)
print(s['why'], s['num_iterations'], s['total_time_ms'])

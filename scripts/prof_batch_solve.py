"""rocprofv3 --kernel-trace workload: ea_batch_solve of 32 x C2 (fp32), default (two concurrent halves)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
dtype = capi.EA_F32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else capi.EA_F64
Ps = []
for sd in range(100, 132):
    cfg = synth.config_c2_twin(seed=sd)
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0); Ps.append(P)
B = capi.Batch(Ps)
if len(sys.argv) > 2: B.set_tuning('solve_streams', int(sys.argv[2]))
q0 = np.tile([1., 0, 0, 0], (32, 1)); t0 = np.zeros((32, 3))
for _ in range(3): B.solve(q0, t0)
t_ = time.perf_counter()
for _ in range(5): q, t, s = B.solve(q0, t0)
print('solve %.3f ms per 32, iterations %s' % ((time.perf_counter() - t_) / 5 * 1e3, sorted(x['num_iterations'] for x in s)))

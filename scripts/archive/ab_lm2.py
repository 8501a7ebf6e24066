"""Same-box A/B of LM solve time between two builds (EA_HIP_LIB) or tuning: C2, 1e5 fp64 / fp32, 32 x C2 batch."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
tag = os.path.basename(os.environ.get('EA_HIP_LIB', 'current'))
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
def solve_time(Ps, reps=40):
    B = capi.Batch(Ps); m = len(Ps)
    Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
    for _ in range(3): q, t, s = B.solve(Q, T)
    best = 1e9
    for r in range(7):
        t_ = time.perf_counter()
        for _ in range(reps): q, t, s = B.solve(Q, T)
        best = min(best, (time.perf_counter() - t_) / reps)
    its = sum(x['num_iterations'] for x in s)
    B.close()
    return best, its, q[0].copy()
def prob(cfg, dtype):
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0); return P
for name, cfgs, dtype in (('C2 f64', [synth.config_c2_twin()], capi.EA_F64), ('lm1e5 f64', [synth.config_c2_twin(seed=7, n_points=100000)], capi.EA_F64),
                          ('lm1e5 f32', [synth.config_c2_twin(seed=7, n_points=100000)], capi.EA_F32),
                          ('32xC2 f64', [synth.config_c2_twin(seed=100 + i) for i in range(32)], capi.EA_F64)):
    Ps = [prob(c, dtype) for c in cfgs]
    best, its, q = solve_time(Ps, 40 if len(Ps) == 1 else 10)
    print('[%s] %-10s solve %.1f us, %d it, %.0f it/s, %.2f us/it | q %s' % (tag, name, best * 1e6, its, its / best, best * 1e6 / (its / len(Ps)), np.array2string(q, precision=12)), flush=True)
    for P in Ps: P.close()

"""A/B of one tuning key on the same box: LM solve time / iterations per second, results compared.
usage: python scripts/ab_tuning.py KEY V0 V1"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
def prob(n, dtype, seed=7):
    cfg = synth.config_c2_twin(seed=seed, n_points=n)
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    return P
for name, n, dtype in (('f64 100k', 100000, capi.EA_F64), ('f64 50k', 50000, capi.EA_F64), ('f64 3k', 3000, capi.EA_F64), ('f32 100k', 100000, capi.EA_F32), ('f32 45k', 45000, capi.EA_F32)):
    P = prob(n, dtype)
    out = {}
    for rep in range(2):
        for v in (v0, v1):
            B = capi.Batch([P]); B.set_tuning(key, v)
            for _ in range(3): q, t, s = B.solve(q0, t0)
            best = 1e9
            for r in range(5):
                t_ = time.perf_counter()
                for _ in range(30): q, t, s = B.solve(q0, t0)
                best = min(best, (time.perf_counter() - t_) / 30)
            prev = out.get(v, (1e9,))
            out[v] = (min(best, prev[0]), s[0]['num_iterations'], q.copy(), t.copy(), s[0]['final_cost'], B.info('num_tiles'))
            B.close()
    a, b = out[v0], out[v1]
    print('%-9s %s=%d: %.1f us/solve (%d it, %.0f it/s) | %s=%d: %.1f us/solve (%d it, %.0f it/s) | rows %d | dq %.1e dt %.1e' % (
        name, key, v0, a[0] * 1e6, a[1], a[1] / a[0], key, v1, b[0] * 1e6, b[1], b[1] / b[0], a[5], np.abs(a[2] - b[2]).max(), np.abs(a[3] - b[3]).max()), flush=True)
    P.close()

"""Persistent one-launch solve vs the (evaluate, step) launch pairs: identical results, timing."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
def make(n, dtype, seed=7):
    cfg = synth.config_c2_twin(seed=seed, n_points=n)
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    return P
for n, dtype in ((100000, capi.EA_F64), (50000, capi.EA_F64), (100000, capi.EA_F32), (3000, capi.EA_F64)):
    P = make(n, dtype)
    res = {}
    for mode in (0, 1):
        B = capi.Batch([P]); B.set_tuning('persistent', mode)
        q, t, s = B.solve(q0, t0)
        used = B.info('last_solve_persistent')
        for _ in range(3): B.solve(q0, t0)
        best = 1e9
        for rep in range(5):
            t_ = time.perf_counter()
            for _ in range(20): B.solve(q0, t0)
            best = min(best, (time.perf_counter() - t_) / 20)
        res[mode] = (q, t, s[0], best, used)
        B.close()
    (qa, ta, sa, ba, ua), (qb, tb, sb, bb, ub) = res[0], res[1]
    same = np.allclose(qa, qb, rtol=0, atol=1e-12) and np.allclose(ta, tb, rtol=0, atol=1e-12) and sa['num_iterations'] == sb['num_iterations'] and np.allclose(sa['it_cost'], sb['it_cost'], rtol=1e-12)
    print('n=%d dtype=%d: loop %.1f us (%d it, persistent=%d) | one launch %.1f us (%d it, persistent=%d) | identical=%s maxrel=%.1e | it/s %.0f -> %.0f' % (
        n, dtype, ba * 1e6, sa['num_iterations'], ua, bb * 1e6, sb['num_iterations'], ub, same, max(np.abs(qa-qb).max(), np.abs(ta-tb).max()), sa["num_iterations"] / ba, sb['num_iterations'] / bb), flush=True)
    if not same:
        print('  dq', np.abs(qa - qb).max(), 'dt', np.abs(ta - tb).max(), sa['why'], sb['why'])
    P.close()
# batch of 4 uneven problems
Ps = [make(n, capi.EA_F64, seed=10 + i) for i, n in enumerate((20000, 5000, 12000, 777))]
out = {}
for mode in (0, 1):
    B = capi.Batch(Ps); B.set_tuning('persistent', mode)
    q, t, s = B.solve(np.tile(q0, (4, 1)), np.tile(t0, (4, 1)))
    out[mode] = (q, t, [x['num_iterations'] for x in s], B.info('last_solve_persistent'))
    B.close()
print('batch of 4: iterations', out[0][2], out[1][2], 'persistent used', out[1][3], 'identical', np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]))

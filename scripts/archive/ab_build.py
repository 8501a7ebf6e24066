"""Same-box A/B of two builds of the library (EA_HIP_LIB=... selects the build) or of tuning keys (k=v arguments):
back-to-back kernel time and step time of the fused evaluation on the bench's workloads, LM solve time at 1e5 points."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
tune = dict(kv.split('=') for kv in sys.argv[1:] if '=' in kv)
only = [a for a in sys.argv[1:] if '=' not in a]
q0 = np.array([1., 0, 0, 0])
tag = os.path.basename(os.environ.get('EA_HIP_LIB', 'current')) + ' ' + ' '.join('%s=%s' % kv for kv in tune.items())


def run(name, cfgs, dtype, loss, tile=None, steps=200, solve=False):
    if only and not any(o in name for o in only):
        return
    Ps = []
    for cfg in cfgs:
        P = capi.Problem(*cfg['K'], dtype=dtype)
        if tile is not None:
            P.set_point_order(tile)
        P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(*loss); Ps.append(P)
    B = capi.Batch(Ps); n = sum(P.num_points for P in Ps); m = len(Ps)
    for k, v in tune.items():
        B.set_tuning(k, int(v))
    Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
    g = B.eval(Q, T)
    best_k, best_s = 1e9, 1e9
    for rep in range(3):
        ms, _ = B.bench_eval(Q, T, 10, steps, kernel_pass=False)
        msk = B.bench_kernel(Q, T, 5, steps)
        best_k, best_s = min(best_k, msk), min(best_s, ms / steps)
    extra = ''
    if solve:
        B.solve(Q, T)
        best = 1e9
        for rep in range(5):
            t_ = time.perf_counter()
            for _ in range(20):
                qs, ts, ss = B.solve(Q, T)
            best = min(best, (time.perf_counter() - t_) / 20)
        extra = ' | solve %.1f us, %d it, %.0f it/s' % (best * 1e6, ss[0]['num_iterations'], sum(s['num_iterations'] for s in ss) / best)
    print('[%s] %-18s ppt %d nt %4d rows %5d | kernel(b2b) %6.2f us step %6.2f us | %.3e evals/s | cost[0] %.12g%s' % (
        tag, name, B.info('points_per_thread'), B.info('threads'), B.info('num_tiles'), best_k * 1e3, best_s * 1e3,
        n / (best_k * 1e-3), g['cost'][0], extra), flush=True)
    B.close()
    for P in Ps:
        P.close()


c2s = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
cauchy, trivial = (capi.LOSS_CAUCHY, 1.0), (capi.LOSS_TRIVIAL, 1.0)
run('32xC2 f32 raster', c2s, capi.EA_F32, cauchy, solve=True)
run('32xC2 f32 tile16', c2s, capi.EA_F32, cauchy, tile=16)
run('32xC2 f64 raster', c2s, capi.EA_F64, cauchy, solve=True)
run('32xC2 f64 tile16', c2s, capi.EA_F64, cauchy, tile=16)
c5 = synth.config_c5()
run('C5 f32', [c5], capi.EA_F32, trivial)
run('C5 f64', [c5], capi.EA_F64, trivial)
run('C2 f64', [synth.config_c2_twin()], capi.EA_F64, cauchy, steps=500, solve=True)
run('lm1e5 f64', [synth.config_c2_twin(seed=7, n_points=100000)], capi.EA_F64, cauchy, steps=500, solve=True)
run('lm1e5 f32', [synth.config_c2_twin(seed=7, n_points=100000)], capi.EA_F32, cauchy, steps=500, solve=True)

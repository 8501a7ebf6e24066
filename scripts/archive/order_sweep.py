"""Does the storage ORDER of the edge points matter?  The caller's raster order (the reference extractor's) against
tiles of T x T pixels (ea_problem_set_point_order): kernel time of the fused evaluation per order and launch shape."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0])

def run(name, cfgs, dtype, loss, Ts, tunings=((-1, -1),), steps=200):
    for T in Ts:
        Ps = []
        for cfg in cfgs:
            P = capi.Problem(*cfg['K'], dtype=dtype); P.set_point_order(T); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(*loss); Ps.append(P)
        B = capi.Batch(Ps); n = sum(P.num_points for P in Ps); m = len(Ps)
        rng = np.random.default_rng(3)
        Q = np.tile(q0, (m, 1)) + 0.01 * rng.standard_normal((m, 4)); Q /= np.linalg.norm(Q, axis=1)[:, None]
        Tt = 0.01 * rng.standard_normal((m, 3))
        for ppt, nt in tunings:
            B.set_tuning('points_per_thread', ppt); B.set_tuning('threads', nt)
            out = B.eval(Q, Tt)
            ms, msk = B.bench_eval(Q, Tt, 10, steps)
            kb = B.bench_kernel(Q, Tt, 20, 400)
            print('%s tile %3d (in effect %3d) ppt %d nt %4d rows %5d | step %.2f us kernel(b2b) %.2f us | %.3e evals/s | cost[0] %.12g' % (
                name, T, Ps[0].point_order, B.info('points_per_thread'), B.info('threads'), B.info('num_tiles'), ms / steps * 1e3, kb * 1e3, n / (ms / steps * 1e-3), out['cost'][0]), flush=True)
        B.close()
        for P in Ps: P.close()

if __name__ == '__main__':
    which = sys.argv[1:] or ['b32', 'b64', 'c5', 'c5rand', 'c3', 'c2']
    c2s = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
    if 'b32' in which: run('32xC2 f32', c2s, capi.EA_F32, (capi.LOSS_CAUCHY, 1.0), (0, 8, 16, 32), ((1, 256), (2, 256)))
    if 'b64' in which: run('32xC2 f64', c2s, capi.EA_F64, (capi.LOSS_CAUCHY, 1.0), (0, 8, 16, 32), ((1, 256), (2, 256)))
    if 'c5' in which:
        c5 = synth.config_c5()
        run('C5 f32', [c5], capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0), (0, 8, 16, 32, 64, -1), ((-1, -1), (2, 1024), (4, 256)))
        run('C5 f64', [c5], capi.EA_F64, (capi.LOSS_TRIVIAL, 1.0), (0, 16), ((-1, -1),))
    if 'c5rand' in which:  # the same cloud handed over in random order (worst case for the caller's order)
        c5r = synth.config_c5(order="random")
        run('C5 f32, points handed over in RANDOM order', [c5r], capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0), (0, -1), ((-1, -1),))
    if 'c3' in which:
        l0 = synth.config_c3_levels()[0]
        run('C3 level 0 (1280x960, 1.4e5 pts) f32', [l0], capi.EA_F32, (capi.LOSS_CAUCHY, 1.0), (0, 16))
    if 'c2' in which: run('C2 f64', [synth.config_c2_twin()], capi.EA_F64, (capi.LOSS_CAUCHY, 1.0), (0, 16))

"""Focused sweep: c5 fp32/fp64 with 1024-thread workgroups and more points per lane."""
import sys, itertools, numpy as np
sys.path.insert(0, '.')
from edge_alignment_amd import capi, synth
q0 = np.array([1.,0,0,0]); t0 = np.zeros(3)
def sweep(name, cfgs, dtype, loss, grid, steps=200):
    Ps = []
    for cfg in cfgs:
        P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(*loss); Ps.append(P)
    B = capi.Batch(Ps); n = sum(P.num_points for P in Ps); m = len(Ps)
    Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
    for ppt, nt in grid:
        B.set_tuning('points_per_thread', ppt); B.set_tuning('threads', nt)
        ms, msk = B.bench_eval(Q, T, 20, steps)
        print('%s ppt %d nt %4d -> rows %5d | step %.2f us kernel(ev) %.2f us | %.3e evals/s' % (
            name, B.info('points_per_thread'), B.info('threads'), B.info('num_tiles'), ms/steps*1e3, msk*1e3, n/(ms/steps*1e-3)), flush=True)
    B.close()
    for P in Ps: P.close()
c5 = synth.config_c5()
sweep('c5 f32', [c5], capi.EA_F32, (capi.LOSS_TRIVIAL,1.0), [(1,256),(2,256),(4,256),(1,1024),(2,1024),(4,1024)])
for n in (200000, 500000, 2000000, 4000000):
    cfg = synth.config_c5(n_points=n)
    sweep('c5-like n=%d f32' % n, [cfg], capi.EA_F32, (capi.LOSS_TRIVIAL,1.0), [(2,256),(4,256),(2,1024),(4,1024)])

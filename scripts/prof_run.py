"""Workload for rocprofv3 runs: a few fused evaluations + LM solves on C2, LM-1e5 and C5."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1.,0,0,0]); t0 = np.zeros(3)
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
def run(cfg, dtype, loss, steps, solve=True, tune=None):
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(*loss)
    B = capi.Batch([P])
    for k, v in (tune or {}).items(): B.set_tuning(k, v)
    ms, _ = B.bench_eval(q0, t0, 5, steps, kernel_pass=False)
    print('ms/step', ms/steps, 'tiles', B.info('num_tiles'), 'ppt', B.info('points_per_thread'))
    if solve:
        q, t, s = P.solve(q0, t0)
        print('solve', s['why'], s['num_iterations'], s['total_time_ms'])
    B.close(); P.close()
if which in ('all', 'c2'):
    run(synth.config_c2_twin(seed=2, n_points=50000), capi.EA_F64, (capi.LOSS_CAUCHY, 1.0), 50)
if which in ('all', 'lm'):
    run(synth.config_c2_twin(seed=7, n_points=100000), capi.EA_F64, (capi.LOSS_CAUCHY, 1.0), 50)
if which in ('all', 'c5'):
    run(synth.config_c5(), capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0), 50)
if which in ('batch32f32', 'batch32f64', 'batchf32', 'batchf64'):   # batchf32 / batchf64: pairs=N (default 32)
    dtype = capi.EA_F32 if which.endswith('f32') else capi.EA_F64
    Ps = []
    tile = [int(kv.split('=')[1]) for kv in sys.argv[2:] if kv.startswith('tile=')]
    pairs = ([int(kv.split('=')[1]) for kv in sys.argv[2:] if kv.startswith('pairs=')] or [32])[0]
    cfgs = [synth.config_c2_twin(seed=100 + i) for i in range(min(pairs, 32))]
    for i in range(pairs):
        cfg = cfgs[i % 32]
        P = capi.Problem(*cfg['K'], dtype=dtype)
        if tile:
            P.set_point_order(tile[0])   # storage order of the points: tiles of this many pixels
        P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    B = capi.Batch(Ps)
    for kv in sys.argv[2:]:
        if kv.startswith('tile=') or kv.startswith('pairs='):
            continue
        k, v = kv.split('='); B.set_tuning(k, int(v))
    ms, _ = B.bench_eval(np.tile(q0, (pairs, 1)), np.tile(t0, (pairs, 1)), 5, 30, kernel_pass=False)
    print('ms/step', ms / 30, 'tiles', B.info('num_tiles'), 'ppt', B.info('points_per_thread'), 'threads', B.info('threads'))

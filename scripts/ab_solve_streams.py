"""The concurrent parts of ea_batch_solve (tuning key "solve_streams"): 32 x C2, fp64 and fp32, interleaved rounds.
usage: python scripts/ab_solve_streams.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch; torch.cuda.init()
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0])
for dtype, name in ((capi.EA_F64, 'f64'), (capi.EA_F32, 'f32')):
    Ps = []
    for i in range(32):
        cb = synth.config_c2_twin(seed=100 + i)
        Pb = capi.Problem(*cb['K'], dtype=dtype); Pb.set_points(cb['xyz']); Pb.set_dt_grid(cb['grid']); Pb.set_loss(capi.LOSS_CAUCHY, 1.0)
        Ps.append(Pb)
    B = capi.Batch(Ps)
    Q = np.tile(q0, (32, 1)); T = np.zeros((32, 3))
    res = {}
    for rnd in range(4):
        for streams in (1, 2, 3):
            B.set_tuning('solve_streams', streams)
            for _ in range(3): B.solve(Q, T)
            ts = []
            for rep in range(6):
                t_ = time.perf_counter()
                for _ in range(5): q, t, s = B.solve(Q, T)
                ts.append((time.perf_counter() - t_) / 5)
            res.setdefault(streams, []).append((min(ts), sorted(ts)[len(ts) // 2]))
    its = sum(x['num_iterations'] for x in s)
    for streams, v in res.items():
        print('batch32 %s solve_streams %d: best %s ms, medians %s ms (%d iterations in all)' % (name, streams, ' '.join('%.3f' % (a * 1e3) for a, _ in v), ' '.join('%.3f' % (b * 1e3) for _, b in v), its), flush=True)
    B.close()
    for P in Ps: P.close()

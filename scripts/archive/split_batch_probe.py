"""32 frame pairs solved by ea_batch_solve as 1..8 concurrent sub-batches (tuning key "solve_streams"): one host
thread pumps all streams; one part's latency-bound LM-step kernel runs under the other parts' evaluations."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth

for dtype, name in ((capi.EA_F32, 'f32'), (capi.EA_F64, 'f64')):
    Ps = []
    for sd in range(100, 132):
        cfg = synth.config_c2_twin(seed=sd)
        P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    q0 = np.tile([1., 0, 0, 0], (32, 1)); t0 = np.zeros((32, 3))
    B = capi.Batch(Ps)
    for streams in (1, 2, 3, 4, 6, 8, -1):
        B.set_tuning('solve_streams', streams)
        for _ in range(3): B.solve(q0, t0)
        best = 1e9
        for rep in range(3):
            t_ = time.perf_counter()
            for _ in range(20): q, t, s = B.solve(q0, t0)
            best = min(best, (time.perf_counter() - t_) / 20)
        iters = sum(x['num_iterations'] for x in s)
        print('%s 32 x C2, solve_streams %2d: %.3f ms per 32 solves, %d LM iterations, %.3e it/s' % (name, streams, best * 1e3, iters, iters / best), flush=True)
    B.close()
    for P in Ps: P.close()

import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
cfg = synth.config_c2_twin(seed=7, n_points=100000)
P = capi.Problem(*cfg['K'], dtype=capi.EA_F64); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
for rep in range(2):
    for ahead in (1, 2, 3, 4, 6):
        for _ in range(5): P.solve(q0, t0, iterations_per_sync=ahead)
        best = 1e9
        for r in range(5):
            t_ = time.perf_counter()
            for _ in range(40): q, t, s = P.solve(q0, t0, iterations_per_sync=ahead)
            best = min(best, (time.perf_counter() - t_) / 40)
        print('ahead %d: %.1f us/solve, %.0f it/s' % (ahead, best * 1e6, s['num_iterations'] / best), flush=True)

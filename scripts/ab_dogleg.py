"""LM and traditional dogleg at C2 (25 iterations allowed), one launch per iteration and as pairs: it/s through ctypes.
usage: python scripts/ab_dogleg.py   (to compare two builds: set capi.LIB_PATH before running the body, as gpu A/Bs do)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch; torch.cuda.init()
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
cfg = synth.config_c2_twin(seed=7, n_points=50000)
P = capi.Problem(*cfg['K'], dtype=capi.EA_F64); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
B = capi.Batch([P])
for strat in (capi.STRATEGY_LM, capi.STRATEGY_DOGLEG):
    for fused in (-1, 0):
        B.set_tuning('fused_iterations', fused)
        for _ in range(5): q, t, s = B.solve(q0, t0, strategy=strat, max_num_iterations=25)
        best = 1e9
        for rep in range(5):
            t_ = time.perf_counter()
            for _ in range(40): q, t, s = B.solve(q0, t0, strategy=strat, max_num_iterations=25)
            best = min(best, (time.perf_counter() - t_) / 40)
        print('strategy %d fused %d: %d iterations, %.1f us per solve, %.0f it/s' % (strat, fused, s[0]['num_iterations'], best * 1e6, s[0]['num_iterations'] / best), flush=True)

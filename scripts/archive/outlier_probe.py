"""Re-measure the one unexplained line of profiles/r01_sweep.txt: `c5 f32 lds 1 ppt 2 nt 256 -> step 465.29 us` (kernel 13.4 us)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
cfg = synth.config_c5()
P = capi.Problem(*cfg['K'], dtype=capi.EA_F32); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_TRIVIAL, 1.0)
B = capi.Batch([P])
for use_lds, ppt, nt in ((1, 2, 256), (1, 2, 256), (0, 2, 256), (1, 1, 256), (1, 4, 256), (1, 2, 1024), (1, 2, 256)):
    B.set_tuning('use_lds', use_lds); B.set_tuning('points_per_thread', ppt); B.set_tuning('threads', nt)
    for rep in range(2):
        ms, msk = B.bench_eval(q0, t0, 10, 100)
        fold = B.bench_fold(5, 100)
        print('c5 f32 lds %d ppt %d nt %4d rows %5d lds_bytes %6d | step %8.2f us kernel(ev) %6.2f us fold(b2b) %6.2f us' % (
            use_lds, B.info('points_per_thread'), B.info('threads'), B.info('num_tiles'), B.info('lds_bytes'), ms / 100 * 1e3, msk * 1e3, fold * 1e3), flush=True)

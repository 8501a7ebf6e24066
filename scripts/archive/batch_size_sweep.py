"""Roofline fraction of the fused evaluation against the number of frame pairs in one launch (C2-shaped pairs: 640x480,
50 000 points): how much of a launch is ramp and tail.  kernel(b2b) = mean over 100 launches executing from the queue."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0])
cfgs = [synth.config_c2_twin(seed=100 + i) for i in range(32)]


def run(m, dtype, tile):
    esz = 4 if dtype == capi.EA_F32 else 8
    Ps = []
    for i in range(m):
        cfg = cfgs[i % 32]
        P = capi.Problem(*cfg['K'], dtype=dtype)
        if tile:
            P.set_point_order(tile)
        P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0); Ps.append(P)
    B = capi.Batch(Ps)
    Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
    B.eval(Q, T)
    best = min(B.bench_kernel(Q, T, 5, 100) for _ in range(3))
    n = sum(P.num_points for P in Ps)
    by = sum(3 * esz * P.num_points + 480 * 640 * esz for P in Ps)
    print('%3d pairs %s %-7s | rows %6d | kernel(b2b) %8.2f us | %.3e evals/s | %6.0f GB/s algorithmic = %.3f of 8 TB/s' % (
        m, 'f32' if esz == 4 else 'f64', 'tile16' if tile else 'raster', B.info('num_tiles'), best * 1e3, n / (best * 1e-3),
        by / (best * 1e-3) / 1e9, by / (best * 1e-3) / 1e9 / 8000.0), flush=True)
    B.close()
    for P in Ps:
        P.close()


for dtype in (capi.EA_F32, capi.EA_F64):
    for tile in (0, 16):
        for m in (1, 4, 8, 16, 32, 64, 128, 256):
            run(m, dtype, tile)

"""Launch shape of the fused evaluation for ONE large problem (C5-shaped: 2048x1536 image, raster and tile order):
workgroup size x points per lane against the number of points.  kernel(b2b) = mean over 100 launches, best of 3."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
full = synth.config_c5()
for dtype in (capi.EA_F32, capi.EA_F64):
    esz = 4 if dtype == capi.EA_F32 else 8
    for n in (125000, 250000, 500000, 1000000):
        for tile in (0, 16, 32):
            P = capi.Problem(*full['K'], dtype=dtype)
            if tile: P.set_point_order(tile)
            P.set_points(full['xyz'][:: (1000000 // n)][:n]); P.set_dt_grid(full['grid']); P.set_loss(capi.LOSS_TRIVIAL, 1.0)
            B = capi.Batch([P])
            B.eval(q0, t0)
            auto = (B.info('points_per_thread'), B.info('num_tiles'))
            row = []
            for nt in (256, 1024):
                for ppt in (1, 2, 4):
                    B.set_tuning("threads", nt); B.set_tuning("points_per_thread", ppt)
                    B.eval(q0, t0)
                    if B.info('points_per_thread') != ppt:
                        row.append("   -  "); continue
                    best = min(B.bench_kernel(q0, t0, 5, 100) for _ in range(3))
                    by = 3 * esz * P.num_points + 2048 * 1536 * esz
                    row.append("%5.2f(%.2f)" % (best * 1e3, by / (best * 1e-3) / 1e9 / 8000.0))
            print("%s n %7d tile %2d | auto ppt %d rows %5d | 256x1 256x2 256x4 1024x1 1024x2 1024x4 us(frac): %s" % (
                "f32" if esz == 4 else "f64", n, tile, auto[0], auto[1], " ".join(row)), flush=True)
            B.close(); P.close()

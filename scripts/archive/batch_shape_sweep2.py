"""fp32 batches: one or two points per lane against the number of pairs (raster and tile order); also the batch solve."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0])
cfgs = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
for tile in (0, 16):
    for m in (4, 8, 12, 16, 24, 32, 64, 128, 256):
        Ps = []
        for i in range(m):
            cfg = cfgs[i % 32]
            P = capi.Problem(*cfg['K'], dtype=capi.EA_F32)
            if tile: P.set_point_order(tile)
            P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0); Ps.append(P)
        B = capi.Batch(Ps)
        Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
        out = []
        for ppt in (1, 2):
            B.set_tuning("points_per_thread", ppt)
            B.eval(Q, T)
            best = min(B.bench_kernel(Q, T, 5, 100) for _ in range(3))
            ts = []
            if m <= 64:
                B.solve(Q, T)
                for _ in range(5):
                    t0 = time.perf_counter(); B.solve(Q, T); ts.append(time.perf_counter() - t0)
            out.append((best * 1e3, min(ts) * 1e3 if ts else float("nan")))
        print("%3d pairs f32 %-6s | kernel ppt1 %7.2f us ppt2 %7.2f us (%+5.1f%%) | batch solve ppt1 %.3f ms ppt2 %.3f ms" % (
            m, "tile16" if tile else "raster", out[0][0], out[1][0], (out[1][0] / out[0][0] - 1) * 100, out[0][1], out[1][1]), flush=True)
        B.close()
        for P in Ps: P.close()

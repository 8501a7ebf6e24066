"""Launch shape of the fused evaluation for batches of C2-shaped pairs: points per lane x workgroup size, tile order.
kernel(b2b) = mean over 100 launches executing from the queue, best of 3."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
q0 = np.array([1., 0, 0, 0])
cfgs = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
for dtype in (capi.EA_F32, capi.EA_F64):
    esz = 4 if dtype == capi.EA_F32 else 8
    for m in (32, 64):
        Ps = []
        for i in range(m):
            cfg = cfgs[i % 32]
            P = capi.Problem(*cfg['K'], dtype=dtype); P.set_point_order(16)
            P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0); Ps.append(P)
        B = capi.Batch(Ps)
        Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
        ref = None
        for nt in (256, 1024):
            for ppt in (1, 2, 4):
                try:
                    B.set_tuning("threads", nt); B.set_tuning("points_per_thread", ppt)
                    g = B.eval(Q, T)
                    best = min(B.bench_kernel(Q, T, 5, 100) for _ in range(3))
                except capi.EAError as e:
                    print(m, nt, ppt, "refused:", e); continue
                by = sum(3 * esz * P.num_points + 480 * 640 * esz for P in Ps)
                if ref is None: ref = g["cost"].copy()
                print('%3d pairs %s | threads %4d ppt %d (got %d x %d) rows %6d | kernel(b2b) %8.2f us | frac %.3f | cost rel diff %.1e' % (
                    m, 'f32' if esz == 4 else 'f64', nt, ppt, B.info('threads') if False else nt, B.info('points_per_thread'), B.info('num_tiles'), best * 1e3,
                    by / (best * 1e-3) / 1e9 / 8000.0, np.abs(g["cost"] - ref).max() / np.abs(ref).max()), flush=True)
        B.close()
        for P in Ps: P.close()

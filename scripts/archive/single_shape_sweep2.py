"""Launch shape for ONE problem: kernel, (evaluate + fold) step under graph replay, and a whole LM solve, per shape.
C5-shaped problems (2048x1536 image) and the 1e5-point LM workload of bench.py."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from edge_alignment_amd import capi, synth
import bench
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
full = synth.config_c5()
lm, lm_dt, _, lm_loss, _ = bench.build_workload("lm1e5", 0)
work = [("lm1e5", lm, lm["xyz"], lm_loss)] + [("c5/%d" % n, full, full['xyz'][:: (1000000 // n)][:n], (capi.LOSS_TRIVIAL, 1.0)) for n in (150000, 200000, 250000, 350000, 500000, 1000000)]
for dtype in (capi.EA_F32, capi.EA_F64):
    for name, cfg, X, loss in work:
        P = capi.Problem(*cfg['K'], dtype=dtype)
        P.set_points(X); P.set_dt_grid(cfg['grid']); P.set_loss(*loss)
        B = capi.Batch([P])
        B.eval(q0, t0)
        auto = (B.info('points_per_thread'), B.info('num_tiles'))
        cells = []
        for nt in (256, 1024):
            for ppt in (1, 2, 4):
                B.set_tuning("threads", nt); B.set_tuning("points_per_thread", ppt)
                B.eval(q0, t0)
                if B.info('points_per_thread') != ppt:
                    continue
                k = min(B.bench_kernel(q0, t0, 5, 100) for _ in range(3)) * 1e3
                B.bench_eval(q0, t0, 0, 5, kernel_pass=False)
                B.bench_capture(200)
                st = min(B.bench_steps(200, host_times=True)[2] for _ in range(3)) / 200 * 1e3
                B.solve(q0[None], t0[None])
                ts = []
                for _ in range(5):
                    a = time.perf_counter(); _, _, ss = B.solve(q0[None], t0[None]); ts.append(time.perf_counter() - a)
                cells.append("%dx%d r%-4d k %5.2f step %5.2f solve %6.1f us/%d it" % (nt, ppt, B.info('num_tiles'), k, st, min(ts) * 1e6, ss[0]["num_iterations"]))
        print("%s %-10s n %7d auto ppt %d rows %4d\n    " % ("f32" if dtype == capi.EA_F32 else "f64", name, P.num_points, auto[0], auto[1]) + "\n    ".join(cells), flush=True)
        B.close(); P.close()

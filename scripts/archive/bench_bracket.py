"""What a K-step timed region costs beyond K x (steady-state step): wall clock around bench_steps(K) for a sweep of K on
the C2 workload, with the library's own split into host enqueue time and wait time, and the torch synchronisation the
bench contract puts around the region."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edge_alignment_amd import capi, synth
cfg = synth.config_c2_twin(seed=2, n_points=50000)
P = capi.Problem(*cfg["K"], dtype=capi.EA_F64); P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
B = capi.Batch([P])
q0, t0 = np.array([1., 0, 0, 0]), np.zeros(3)
B.bench_eval(q0, t0, 0, 50, kernel_pass=False)
use_graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
for K in (1, 2, 5, 10, 20, 50, 100, 500, 2000):
    if use_graph:
        B.bench_capture(K)
    best = None
    for rep in range(7):
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        us = B.bench_steps(K, host_times=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        row = ((t2 - t_) * 1e6, (t1 - t_) * 1e6, us[0], us[1], (t2 - t1) * 1e6)
        best = row if best is None or row[0] < best[0] else best
    print("K %5d | bracket %8.1f us = %.2f us/step | call %8.1f (enqueue %7.1f wait %6.1f) torch.sync %5.1f" % (
        K, best[0], best[0] / K, best[1], best[2], best[3], best[4]), flush=True)

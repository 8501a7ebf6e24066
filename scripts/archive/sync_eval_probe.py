"""Wall clock of the synchronous entry points: ea_batch_eval (one C2 problem, a batch of 32), ea_eval_rows into host arrays,
ea_solve_sharded through the host hook on one rank.   python scripts/sync_eval_probe.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth, dist as ead
tag = os.path.basename(os.environ.get("EA_HIP_LIB", "current"))
cfg = synth.config_c2_twin(seed=2, n_points=50000)
P = capi.Problem(*cfg["K"], dtype=capi.EA_F64); P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
B = capi.Batch([P])
q0, t0 = np.array([1., 0, 0, 0]), np.zeros(3)
g0 = B.eval(q0, t0)
for rep in range(3):
    ts = []
    for _ in range(300):
        t_ = time.perf_counter(); g = B.eval(q0, t0); ts.append(time.perf_counter() - t_)
    assert np.array_equal(g["JtJ"], g0["JtJ"])
    print("[%s] ea_batch_eval, one C2 problem: median %.1f us, min %.1f us" % (tag, np.median(ts) * 1e6, min(ts) * 1e6), flush=True)
cfg2 = synth.config_c2_twin(seed=7, n_points=100000)
P2 = capi.Problem(*cfg2["K"], dtype=capi.EA_F64); P2.set_points(cfg2["xyz"]); P2.set_dt_grid(cfg2["grid"]); P2.set_loss(capi.LOSS_CAUCHY, 1.0)
ar = ead.make_allreduce(1, device="cpu")
P2.solve_sharded(q0, t0, ar)
for rep in range(3):
    t_ = time.perf_counter(); its = 0
    for _ in range(10):
        q, t, s = P2.solve_sharded(q0, t0, ar); its += s["num_iterations"]
    el = time.perf_counter() - t_
    print("[%s] ea_solve_sharded (host hook, one rank), 1e5 points: %.1f us per solve, %.0f it/s" % (tag, el / 10 * 1e6, its / el), flush=True)

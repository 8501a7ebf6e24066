"""Latency of one ea_batch_eval call (the reference's problem.Evaluate) with the pose constants read in place from pinned host
memory (default for <= 4 problems) against uploaded first (tuning key "zero_copy_poses" 0).  usage: python scripts/ab_zero_copy.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth
q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
for name, n, dtype in (("c2_f64", 50000, capi.EA_F64), ("1482_f64", 1482, capi.EA_F64), ("c2_f32", 50000, capi.EA_F32)):
    cfg = synth.config_c2_twin(seed=7, n_points=n)
    P = capi.Problem(*cfg["K"], dtype=dtype); P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    B = capi.Batch([P])
    res, ref = {}, {}
    for rnd in range(3):
        for zc in (1, 0):
            B.set_tuning("zero_copy_poses", -1 if zc else 0)
            for _ in range(20):
                e = B.eval(q0, t0)
            best = 1e9
            for rep in range(5):
                t_ = time.perf_counter()
                for _ in range(200):
                    e = B.eval(q0, t0)
                best = min(best, (time.perf_counter() - t_) / 200)
            res.setdefault(zc, []).append(best * 1e6); ref[zc] = e
    assert all(np.array_equal(ref[1][k], ref[0][k]) for k in ("cost", "JtJ", "Jtr"))
    print("%-9s one ea_batch_eval call: %.1f us with the poses read in place, %.1f us uploaded first (same bits)" % (name, sorted(res[1])[1], sorted(res[0])[1]), flush=True)
    B.close(); P.close()

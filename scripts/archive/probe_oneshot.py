"""One-shot timed regions as bench.py brackets them (device sync, clock, ONE ea_batch_eval_resident_poses call of K = 20 poses,
device sync, clock), ten in a row with different idle gaps in front: is the region's ~46 us the call or the cold start?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from edge_alignment_amd import capi, synth
import bench
cfg = synth.config_c2_twin(seed=2, n_points=50000)
P = capi.Problem(*cfg["K"], dtype=capi.EA_F64); P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
B = capi.Batch([P])
Q, T = bench.step_poses(20, 7)
for poll in (0, 1):
    B.set_tuning("poll_results", poll)
    B.set_poses(Q, T)
    out = B.eval_resident_poses()
    for gap_ms in (0.0, 0.1, 1.0, 10.0, 100.0):
        ts = []
        for _ in range(10):
            B.eval_resident_poses(out=out); B.eval_resident_poses(out=out)
            torch.cuda.synchronize()
            if gap_ms:
                time.sleep(gap_ms * 1e-3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            B.eval_resident_poses(out=out)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        print("poll %d idle gap %6.1f ms: one-shot region %s us (median %.1f)" % (poll, gap_ms, " ".join("%.1f" % x for x in ts), float(np.median(ts))), flush=True)
B.close(); P.close()

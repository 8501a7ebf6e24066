"""Host-side cost of the synchronous evaluation calls on C2: polling the done flag the last fold workgroup raises in pinned
memory against waiting for the stream's completion signal (tuning key "poll_results"), with and without a device
synchronisation behind the call (bench.py's bracket ends on one).  Wall clock, mean of 300 calls, three interleaved rounds.
  python scripts/ab_poll.py > profiles/r03_ab_poll.txt"""
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
q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
for K in (1, 20, 200):
    Q, T = bench.step_poses(K, 7)
    res = {}
    for rnd in range(3):
        for poll in (1, 0):
            B.set_tuning("poll_results", poll)
            B.set_poses(Q, T)
            out = B.eval_resident_poses()
            for dev_sync in (0, 1):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(300):
                    B.eval_resident_poses(out=out)
                    if dev_sync:
                        torch.cuda.synchronize()
                res.setdefault((poll, dev_sync), []).append((time.perf_counter() - t) / 300 * 1e6)
            if K == 1:
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(300):
                    B.eval(q0, t0)
                res.setdefault((poll, "ea_batch_eval"), []).append((time.perf_counter() - t) / 300 * 1e6)
    for k, v in sorted(res.items(), key=str):
        print("K %3d  %-26s %-34s %7.2f us per call  (rounds %s)" % (K, "poll the done flag" if k[0] else "hipStreamSynchronize", "ea_batch_eval (one pose)" if k[1] == "ea_batch_eval" else
              ("ea_batch_eval_resident_poses" + (" + torch.cuda.synchronize" if k[1] else "")), min(v), " ".join("%.2f" % x for x in v)))
B.close(); P.close()

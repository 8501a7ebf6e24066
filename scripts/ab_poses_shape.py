"""Launch shape of the pose-batched evaluation (ea_batch_eval_resident_poses): points per lane x workgroup size, C2 and C5,
K = 20 and K = 2000 -- ms per run of the timed call's launches (events on the library's stream, best of 3 x 5 runs).
  python scripts/ab_poses_shape.py > profiles/r03_ab_poses_shape.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def run(name, cfg, dtype, loss, shapes):
    P = capi.Problem(*cfg["K"], dtype=dtype)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(*loss)
    B = capi.Batch([P])
    n = P.num_points
    for K in (20, 2000):
        Q, T = bench.step_poses(K, 1000)
        for ppt, nt in shapes:
            B.set_tuning("points_per_thread", ppt); B.set_tuning("threads", nt)
            try:
                B.set_poses(Q, T)
                ms, nl = min(B.bench_resident_poses(5) for _ in range(3))
                mse, _ = min(B.bench_resident_poses(5, evaluations_only=True) for _ in range(3))
                print("%-8s K %4d ppt %d nt %4d (effective ppt %d nt %d): %8.2f us per run, %6.3f us per evaluation, evaluation launches %2d x %8.2f us, tiles %d"
                      % (name, K, ppt, nt, B.info("points_per_thread"), B.info("threads"), ms * 1e3, ms * 1e3 / K, nl, mse * 1e3 / nl, B.info("num_tiles")), flush=True)
            except capi.EAError as e:
                print(name, K, ppt, nt, "error", e)
    B.close(); P.close()

run("C2 f64", synth.config_c2_twin(seed=2, n_points=50000), capi.EA_F64, (capi.LOSS_CAUCHY, 1.0), [(1, 256), (2, 256), (1, 1024)])
run("C2 f32", synth.config_c2_twin(seed=2, n_points=50000), capi.EA_F32, (capi.LOSS_CAUCHY, 1.0), [(1, 256), (2, 256), (4, 256), (1, 1024), (2, 1024)])
run("C5 f32", synth.config_c5(), capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0), [(1, 256), (2, 256), (4, 256), (1, 1024), (2, 1024), (4, 1024)])

"""Which launch shape should a single-problem solve take now that 256-thread shapes can run one launch per LM iteration?
For sizes around the thresholds of batch_build: the automatic shape in pair form against 256 x {1, 2, 4} in fused form
(and `iterations_per_sync` 2 / 3 / 4 for the fused form at C2).   usage: python scripts/ab_fused_shapes.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth  # noqa: E402

q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)


def timed(B, **opts):
    for _ in range(3):
        q, t, s = B.solve(q0, t0, **opts)
    best = 1e9
    for rep in range(5):
        t_ = time.perf_counter()
        for _ in range(20):
            q, t, s = B.solve(q0, t0, **opts)
        best = min(best, (time.perf_counter() - t_) / 20)
    return best * 1e6, s[0]["num_iterations"], B.info("fused_iterations"), B.info("threads"), B.info("points_per_thread")


for n, dtype, name in ((60000, capi.EA_F64, "f64"), (70000, capi.EA_F64, "f64"), (79000, capi.EA_F64, "f64"), (120000, capi.EA_F64, "f64"),
                       (70000, capi.EA_F32, "f32"), (140000, capi.EA_F32, "f32"), (200000, capi.EA_F32, "f32"), (250000, capi.EA_F32, "f32")):
    cfg = synth.config_c2_twin(seed=7, n_points=n)
    P = capi.Problem(*cfg["K"], dtype=dtype)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    B = capi.Batch([P])
    rows = []
    B.set_tuning("fused_iterations", 0)
    us, its, fused, nt, ppt = timed(B)
    rows.append("pairs auto %dx%d: %.1f us/it" % (nt, ppt, us / its))
    B.set_tuning("fused_iterations", -1)
    for ppt_try in ((1, 2) if dtype == capi.EA_F64 else (1, 2, 4)):
        B.set_tuning("threads", 256); B.set_tuning("points_per_thread", ppt_try)
        us, its, fused, nt, ppt = timed(B)
        rows.append("%s %dx%d: %.1f us/it" % ("fused" if fused else "pairs", nt, ppt, us / its))
    print("%7d %s | " % (n, name) + " | ".join(rows), flush=True)
    B.close(); P.close()

cfg = synth.config_c2_twin()
P = capi.Problem(*cfg["K"], dtype=capi.EA_F64)
P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
B = capi.Batch([P])
for ahead in (1, 2, 3, 4, 6):
    us, its, fused, nt, ppt = timed(B, iterations_per_sync=ahead)
    print("C2 fused=%d ahead %d: %.1f us per solve, %.2f us/it" % (fused, ahead, us, us / its), flush=True)

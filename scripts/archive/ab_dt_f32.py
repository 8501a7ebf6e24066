"""Same-box A/B of the float32-stored distance transform under fp64 kernels (tuning key "dt_f32": -1 = mirror when every
term has one, 0 = fp64 image), interleaved rounds: evaluation kernel back to back, pipelined step, materialised rows,
batch solve.  Results are bit-identical by construction (tests/test_gpu_dt_f32.py); this measures time only.

  python scripts/ab_dt_f32.py > profiles/r03_ab_dt_f32.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (torch first: one HIP runtime)
if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth

HBM = 8000.0


def build(cfgs, tile=None):
    Ps = []
    for c in cfgs:
        P = capi.Problem(*c["K"], dtype=capi.EA_F64)
        if tile is not None:
            P.set_point_order(tile)
        P.set_points(c["xyz"]); P.set_dt_grid(c["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    return Ps, capi.Batch(Ps)


def measure(name, cfgs, tile=None, rounds=3, solve=False):
    Ps, B = build(cfgs, tile)
    m = len(Ps)
    Q, T = np.tile([1.0, 0, 0, 0], (m, 1)), np.zeros((m, 3))
    npts = sum(P.num_points for P in Ps)
    by8 = sum(3 * 8 * P.num_points + c["image"].shape[0] * c["image"].shape[1] * 8 for P, c in zip(Ps, cfgs))
    res = {0: [], 1: []}
    for r in range(rounds):
        for mode in (1, 0):
            B.set_tuning("dt_f32", -1 if mode else 0)
            B.eval(Q, T)
            assert B.info("dt_f32") == mode
            k = B.bench_kernel(Q, T, 10, 200) * 1e3
            try:
                B.bench_capture_pipelined(100)
                step = min(B.bench_steps(100, host_times=True)[2] for _ in range(3)) / 100 * 1e3
            except capi.EAError:
                step = float("nan")
            rows = min(B.bench_rows(Q, T, 5, 100, corrected=True, layout=0, mode=1) for _ in range(2)) * 1e3
            sv = float("nan")
            if solve:
                B.solve(Q, T)
                t0 = time.perf_counter()
                for _ in range(5):
                    B.solve(Q, T)
                sv = (time.perf_counter() - t0) / 5 * 1e3
            res[mode].append((k, step, rows, sv))
    for mode in (0, 1):
        a = np.array(res[mode])
        best = np.nanmin(a, axis=0)
        print("%-34s %-9s kernel %7.2f us (frac of 8 TB/s at 8-byte texels %.3f)  pipelined step %7.2f us  rows kernel %7.2f us  solve %6.3f ms   [rounds: %s]"
              % (name, "fp32 img" if mode else "fp64 img", best[0], by8 / (best[0] * 1e-6) / 1e9 / HBM, best[1], best[2], best[3],
                 " ".join("%.2f" % x for x in a[:, 0])))
    k0, k1 = min(x[0] for x in res[0]), min(x[0] for x in res[1])
    print("%-34s kernel fp32-image / fp64-image = %.3f" % (name, k1 / k0))
    B.close()
    for P in Ps:
        P.close()


def main():
    print("# A/B of tuning key dt_f32 on one box, %d interleaved rounds, best of rounds; kernel = back-to-back evaluation launches" % 3)
    measure("C2 (5e4 pts, 640x480)", [synth.config_c2_twin(seed=2, n_points=50000)])
    measure("1e5 pts, 640x480", [synth.config_c2_twin(seed=7, n_points=100000)])
    batch = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
    measure("32 x C2 raster", batch, solve=True)
    measure("32 x C2 tile16", batch, tile=16)
    measure("64 x C2 tile16", batch * 2, tile=16)
    measure("128 x C2 tile16 (from HBM)", batch * 4, tile=16)


if __name__ == "__main__":
    main()

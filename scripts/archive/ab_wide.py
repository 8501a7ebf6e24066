"""Cost and effect of ea_batch_set_tuning("wide_accumulate", 1) (fp32 kernels summing in fp64 from the lane's sum on) on
the fp32 workloads of bench.py: evaluation kernel time (back to back, one event pair around 200 launches, interleaved
with the default, best of 5), the relative error of the sums against the fp64 kernel on the same data, and the LM solve.
Output -> profiles/r02_ab_wide_accumulate.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (PyTorch first: profiles/README.md)

from edge_alignment_amd import capi, synth  # noqa: E402
import bench  # noqa: E402


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


def batch_of(problems, dtype, tile=None):
    Ps = []
    for c in problems:
        P = capi.Problem(*c["K"], dtype=dtype, device=0)
        if tile is not None:
            P.set_point_order(tile)
        P.set_points(c["xyz"]); P.set_dt_grid(c["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    return capi.Batch(Ps), Ps


def main():
    torch.zeros(1, device="cuda:0")
    c5 = bench.build_workload("c5", 0)[0]
    cases = [("c5 fp32 1e6 points", [c5], 16),
             ("32 x c2 fp32", [synth.config_c2_twin(seed=100 + i) for i in range(32)], None),
             ("c2 fp32 50000 points", [synth.config_c2_twin(seed=2, n_points=50000)], None),
             ("1e5 points fp32 (LM workload)", [synth.config_c2_twin(seed=3, n_points=100000)], None)]
    for name, problems, tile in cases:
        m = len(problems)
        Q = np.tile(np.array([1.0, 0, 0, 0]), (m, 1)); T = np.zeros((m, 3))
        B64, P64 = batch_of(problems, capi.EA_F64, tile)
        ref = B64.eval(Q, T)
        B64.close()
        for P in P64:
            P.close()
        B, Ps = batch_of(problems, capi.EA_F32, tile)
        times = {0: [], 1: []}
        errs = {}
        for rep in range(5):
            for w in (0, 1):
                B.set_tuning("wide_accumulate", w)
                times[w].append(B.bench_kernel(Q, T, 5, 200) * 1e3)
                if rep == 0:
                    g = B.eval(Q, T)
                    errs[w] = (rel(g["cost"], ref["cost"]), rel(g["JtJ"], ref["JtJ"]), rel(g["Jtr"], ref["Jtr"]))
        solve = {}
        for w in (0, 1):
            B.set_tuning("wide_accumulate", w)
            B.solve(Q, T)
            ts = time.perf_counter()
            for _ in range(5):
                q, t, s = B.solve(Q, T)
            solve[w] = ((time.perf_counter() - ts) / 5 * 1e3, sum(x["num_iterations"] for x in s), q, t)
        dq = max(synth.rotation_angle_between(solve[0][2][i], solve[1][2][i]) for i in range(m))
        dt = float(np.abs(solve[0][3] - solve[1][3]).max())
        print("%-32s shape ppt %d nt %d | kernel default %7.2f us wide %7.2f us (%+5.1f %%) | sums vs fp64 kernel (cost JtJ Jtr): default %.1e %.1e %.1e wide %.1e %.1e %.1e"
              " | solve default %.3f ms (%d it) wide %.3f ms (%d it), poses differ by %.1e rad %.1e m"
              % (name, B.info("points_per_thread"), B.info("threads"), min(times[0]), min(times[1]),
                 (min(times[1]) / min(times[0]) - 1) * 100, *errs[0], *errs[1], solve[0][0], solve[0][1], solve[1][0], solve[1][1], dq, dt), flush=True)
        B.close()
        for P in Ps:
            P.close()


if __name__ == "__main__":
    main()

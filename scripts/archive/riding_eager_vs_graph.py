"""What bench.py's timed bracket costs for a K-step region of the C2 workload when the K riding-fold launches are
(a) replayed from a hipGraph (ea_batch_bench_capture_pipelined + ea_batch_bench_steps) or (b) enqueued launch by launch
(ea_batch_bench_steps_riding).  Same bracket as bench.py: torch.cuda.synchronize(); t0; call; torch.cuda.synchronize(); t1.
Interleaved, median and minimum of `reps` regions per form.  Output -> profiles/r02_riding_eager_vs_graph.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (PyTorch first: profiles/README.md)

from edge_alignment_amd import capi  # noqa: E402
import bench  # noqa: E402


def main():
    torch.cuda.init()
    torch.zeros(1, device="cuda:0")
    reps = int(os.environ.get("REPS", "40"))
    for workload in ("c2", "c5"):
        cfg, dtype, tag, loss, desc = bench.build_workload(workload, 0)
        P = capi.Problem(*cfg["K"], dtype=dtype, device=0)
        P.set_points(cfg["xyz"])
        P.set_dt_grid(cfg["grid"])
        P.set_loss(*loss)
        B = capi.Batch([P])
        q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
        B.bench_eval(q0, t0, 0, 5, kernel_pass=False)
        want = B.eval(q0, t0)
        B.bench_eval(q0, t0, 0, 1, kernel_pass=False)
        print("# %s: %s, %d points" % (workload, desc, P.num_points))
        for K in (5, 20, 50, 100, 500, 2000):
            B.bench_capture_pipelined(K)
            B.bench_steps(K, riding=True)   # warm
            res = {"graph": [], "eager": []}
            split = {"graph": [], "eager": []}
            for r in range(reps):
                for form in ("graph", "eager"):
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    B.bench_steps(K, riding=(form == "eager"))
                    torch.cuda.synchronize()
                    res[form].append((time.perf_counter() - ts) * 1e6)
                    us = B.bench_steps(K, host_times=True, riding=(form == "eager"))
                    split[form].append(us.copy())
            got = B.bench_result()
            assert np.allclose(got["cost"], want["cost"], rtol=1e-12), (got["cost"], want["cost"])
            line = "K %5d |" % K
            for form in ("graph", "eager"):
                a = np.array(res[form])
                sp = np.median(np.array(split[form]), axis=0)
                line += " %s bracket median %8.1f us min %8.1f us = %6.2f us/step (enqueue %7.1f wait %6.1f events %8.1f us) |" % (
                    form, np.median(a), a.min(), np.median(a) / K, sp[0], sp[1], sp[2] * 1e3)
            print(line, flush=True)
        B.close()
        P.close()


if __name__ == "__main__":
    main()

"""Same-box A/B of the two forms of ea_solve's loop: one launch per LM iteration (ea_lm_iter_kernel, tuning key
"fused_iterations" -1) against (evaluate, step) pairs (0).  Interleaved rounds in one process, median of the rounds' best;
the iterates are bit-identical (tests/test_gpu_fused_iterations.py), so both forms run the same number of iterations.
usage: python scripts/ab_fused_iterations.py [rounds]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402  (first: one HIP runtime in the process)

if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
cases = []
for name, n, dtype in (("c2_5e4_f64", 50000, capi.EA_F64), ("1e5_f64", 100000, capi.EA_F64), ("1e5_f32", 100000, capi.EA_F32),
                       ("c1_1482_f64", 1482, capi.EA_F64), ("2e4_f64", 20000, capi.EA_F64)):
    cfg = synth.config_c2_twin(seed=7, n_points=n)
    P = capi.Problem(*cfg["K"], dtype=dtype)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    cases.append((name, P, capi.Batch([P])))

out = {}
for name, P, B in cases:
    res = {1: [], 0: []}
    its = {}
    for r in range(rounds):
        for fused in (1, 0):
            B.set_tuning("fused_iterations", -1 if fused else 0)
            for _ in range(3):
                q, t, s = B.solve(q0, t0)
            best = 1e9
            for rep in range(4):
                t_ = time.perf_counter()
                for _ in range(30):
                    q, t, s = B.solve(q0, t0)
                best = min(best, (time.perf_counter() - t_) / 30)
            assert B.info("fused_iterations") == fused, (name, fused)
            its[fused] = s[0]["num_iterations"]
            res[fused].append(best)
    assert its[0] == its[1]
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    out[name] = {"iterations": its[1], "pairs_solve_us": med[0] * 1e6, "fused_solve_us": med[1] * 1e6,
                 "pairs_it_per_s": its[0] / med[0], "fused_it_per_s": its[1] / med[1],
                 "pairs_us_per_it": med[0] * 1e6 / its[0], "fused_us_per_it": med[1] * 1e6 / its[1],
                 "speedup": med[0] / med[1]}
    print(name, json.dumps({k: round(v, 3) for k, v in out[name].items()}), flush=True)
print(json.dumps(out))

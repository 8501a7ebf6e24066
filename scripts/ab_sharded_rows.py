"""One-rank A/B of ea_solve_sharded_comm's two forms on a one-rank RCCL communicator: one launch per iteration with the partial
rows all-reduced in place (default) against evaluation -> fold -> all-reduce(32 sums) -> step (EA_SHARDED_ROWS=0), with the
unsharded ea_solve beside them.  usage: python scripts/ab_sharded_rows.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth  # noqa: E402

q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)
comm = capi.Comm(capi.comm_unique_id(), 1, 0, device=0)
out = {}
for name, n, dtype in (("1e5_f64", 100000, capi.EA_F64), ("c2_5e4_f64", 50000, capi.EA_F64), ("1e5_f32", 100000, capi.EA_F32)):
    cfg = synth.config_c2_twin(seed=7, n_points=n)
    P = capi.Problem(*cfg["K"], dtype=dtype)
    P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    res = {}
    for rnd in range(3):
        for form in ("rows", "sums", "unsharded"):
            os.environ["EA_SHARDED_ROWS"] = "0" if form == "sums" else "1"
            run = (lambda: P.solve(q0, t0)) if form == "unsharded" else (lambda: P.solve_sharded_comm(q0, t0, comm))
            for _ in range(3):
                q, t, s = run()
            best = 1e9
            for rep in range(4):
                t_ = time.perf_counter()
                for _ in range(30):
                    q, t, s = run()
                best = min(best, (time.perf_counter() - t_) / 30)
            res.setdefault(form, []).append(s["num_iterations"] / best)
    out[name] = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(name, {k: round(v) for k, v in out[name].items()}, "it/s", flush=True)
    P.close()
print(json.dumps(out))
comm.close()

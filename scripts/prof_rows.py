"""Workload for rocprofv3 runs over the materialised-mode kernel: `launches` of ea_eval_rows_kernel on C5 (fp32), C2 (fp64)
or the 32 x C2 batch.   python3 scripts/prof_rows.py c5|c2|batch32f32|batch32f64 [layout] [mode]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if which == "c5":
    cfgs, dtype, loss = [synth.config_c5()], capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0)
elif which == "c2":
    cfgs, dtype, loss = [synth.config_c2_twin(seed=2, n_points=50000)], capi.EA_F64, (capi.LOSS_CAUCHY, 1.0)
else:
    cfgs, loss = [synth.config_c2_twin(seed=100 + i) for i in range(32)], (capi.LOSS_CAUCHY, 1.0)
    dtype = capi.EA_F32 if which.endswith("f32") else capi.EA_F64
Ps = []
for c in cfgs:
    P = capi.Problem(*c["K"], dtype=dtype); P.set_points(c["xyz"]); P.set_dt_grid(c["grid"]); P.set_loss(*loss)
    Ps.append(P)
B = capi.Batch(Ps)
m = len(Ps)
q, t = np.tile(np.array([1.0, 0, 0, 0]), (m, 1)), np.zeros((m, 3))
ms = B.bench_rows(q, t, 3, 30, corrected=True, layout=layout, mode=mode)
print("ms/launch", ms, "rows", int(B.row_offsets()[-1]))

"""Two builds of the library on one box, alternating processes: kernel time per evaluation of the pose-batched launches (C5 fp32,
C2 fp32, C2 fp64).  usage: python scripts/ab_two_builds_poses.py libA.so libB.so"""
import sys, os, json, subprocess
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, '.')
import torch; torch.cuda.init()
from edge_alignment_amd import capi, synth
capi.LIB_PATH = sys.argv[1]
import bench
out = {}
def run(name, cfg, dtype, loss, K):
    P = capi.Problem(*cfg["K"], dtype=dtype); P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(*loss)
    B = capi.Batch([P])
    Q, T = bench.step_poses(K, 1000)
    B.set_poses(Q, T)
    ms, nl = min(B.bench_resident_poses(5, evaluations_only=True) for _ in range(4))
    out[name] = ms * 1e3 / K
    B.close(); P.close()
run("c5_f32_us_per_eval", synth.config_c5(), capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0), 400)
run("c2_f32_us_per_eval", synth.config_c2_twin(seed=2, n_points=50000), capi.EA_F32, (capi.LOSS_CAUCHY, 1.0), 2000)
run("c2_f64_us_per_eval", synth.config_c2_twin(seed=2, n_points=50000), capi.EA_F64, (capi.LOSS_CAUCHY, 1.0), 2000)
import json; print(json.dumps(out))
'''
libs = sys.argv[1:3]
res = {l: [] for l in libs}
for r in range(3):
    for l in libs:
        o = subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(l)], capture_output=True, text=True)
        if o.returncode != 0:
            print(l, 'FAILED', o.stderr[-1500:]); sys.exit(1)
        res[l].append(json.loads(o.stdout.strip().splitlines()[-1]))
for l in libs:
    print(os.path.basename(l), {k: round(sorted(x[k] for x in res[l])[1], 4) for k in res[l][0]})

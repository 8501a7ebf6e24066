"""A/B of two builds of the library on the same box, alternating processes: LM iterations/s of the 1e5-point solve in both forms
of the loop (one launch per iteration / pairs), the 32-pair batch solve (pairs on two streams), C2 evaluation step.
usage: python scripts/ab_lm_forms.py libA.so libB.so [rounds]"""
import subprocess, sys, json, os
CHILD = r'''
import sys, time, json, numpy as np
sys.path.insert(0, '.')
import torch
torch.cuda.init()
from edge_alignment_amd import capi, synth
capi.LIB_PATH = sys.argv[1]
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
out = {}
for name, n in (('1e5', 100000), ('c2', 50000)):
    cfg = synth.config_c2_twin(seed=7, n_points=n)
    P = capi.Problem(*cfg['K'], dtype=capi.EA_F64); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    B = capi.Batch([P])
    for form, val in (('fused', -1), ('pairs', 0)):
        try:
            B.set_tuning('fused_iterations', val)
        except Exception:
            if form == 'fused': continue
        for _ in range(5): q, t, s = B.solve(q0, t0)
        best = 1e9
        for rep in range(5):
            t_ = time.perf_counter()
            for _ in range(40): q, t, s = B.solve(q0, t0)
            best = min(best, (time.perf_counter() - t_) / 40)
        out['%s_%s_it_per_s' % (name, form)] = s[0]['num_iterations'] / best
Ps = []
for i in range(32):
    cb = synth.config_c2_twin(seed=100 + i)
    Pb = capi.Problem(*cb['K'], dtype=capi.EA_F64); Pb.set_points(cb['xyz']); Pb.set_dt_grid(cb['grid']); Pb.set_loss(capi.LOSS_CAUCHY, 1.0)
    Ps.append(Pb)
Bb = capi.Batch(Ps)
Q = np.tile(q0, (32, 1)); T = np.zeros((32, 3))
for _ in range(3): Bb.solve(Q, T)
best = 1e9
for rep in range(4):
    t_ = time.perf_counter()
    for _ in range(5): q, t, s = Bb.solve(Q, T)
    best = min(best, (time.perf_counter() - t_) / 5)
out['batch32_f64_solve_ms'] = best * 1e3
out['batch32_f64_it_per_s'] = sum(x['num_iterations'] for x in s) / best
print(json.dumps(out))
'''
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        o = subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(l)], capture_output=True, text=True)
        if o.returncode != 0:
            print(l, 'FAILED', o.stderr[-2000:]); sys.exit(1)
        res[l].append(json.loads(o.stdout.strip().splitlines()[-1]))
for l in libs:
    keys = res[l][0].keys()
    print(os.path.basename(l), {k: round(float(sorted(x[k] for x in res[l])[len(res[l]) // 2]), 2) for k in keys})

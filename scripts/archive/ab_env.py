"""A/B timing of two builds of the library on the same box: LM iterations/s at 1e5 points and the C2 / C5 step.
usage: python scripts/ab_env.py KEY=VALUE [rounds]   (the shipped library with and without the environment variable; each measurement in its own process, alternating)"""
import subprocess, sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, json, numpy as np
sys.path.insert(0, '.')
from edge_alignment_amd import capi, synth
capi.LIB_PATH = sys.argv[1]
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
cfg = synth.config_c2_twin(seed=7, n_points=100000)
P = capi.Problem(*cfg['K'], dtype=capi.EA_F64); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
for _ in range(5): q, t, s = P.solve(q0, t0)
best = 1e9
for rep in range(5):
    t_ = time.perf_counter()
    for _ in range(40): q, t, s = P.solve(q0, t0)
    best = min(best, (time.perf_counter() - t_) / 40)
its = s['num_iterations']
out = {'lm_it_per_s': its / best, 'solve_us': best * 1e6, 'iters': its}
c2 = synth.config_c2_twin()
P2 = capi.Problem(*c2['K'], dtype=capi.EA_F64); P2.set_points(c2['xyz']); P2.set_dt_grid(c2['grid']); P2.set_loss(capi.LOSS_CAUCHY, 1.0)
B = capi.Batch([P2])
ms = min(B.bench_eval(q0, t0, 20, 300, kernel_pass=False)[0] for _ in range(3))
out['c2_step_us'] = ms / 300 * 1e3
c5 = synth.config_c5()
P5 = capi.Problem(*c5['K'], dtype=capi.EA_F32); P5.set_points(c5['xyz']); P5.set_dt_grid(c5['grid']); P5.set_loss(capi.LOSS_TRIVIAL, 1.0)
B5 = capi.Batch([P5])
ms = min(B5.bench_eval(q0, t0, 20, 200, kernel_pass=False)[0] for _ in range(3))
out['c5_step_us'] = ms / 200 * 1e3
Ps = []
for i in range(32):
    cb = synth.config_c2_twin(seed=100 + i)
    Pb = capi.Problem(*cb['K'], dtype=capi.EA_F32); Pb.set_points(cb['xyz']); Pb.set_dt_grid(cb['grid']); Pb.set_loss(capi.LOSS_CAUCHY, 1.0)
    Ps.append(Pb)
Bb = capi.Batch(Ps)
ms = min(Bb.bench_eval(np.tile(q0, (32, 1)), np.tile(t0, (32, 1)), 10, 100, kernel_pass=False)[0] for _ in range(3))
out['batch32_f32_step_us'] = ms / 100 * 1e3
Ps64 = []
for i in range(32):
    cb = synth.config_c2_twin(seed=100 + i)
    Pb = capi.Problem(*cb['K'], dtype=capi.EA_F64); Pb.set_points(cb['xyz']); Pb.set_dt_grid(cb['grid']); Pb.set_loss(capi.LOSS_CAUCHY, 1.0)
    Ps64.append(Pb)
Bb64 = capi.Batch(Ps64)
ms = min(Bb64.bench_eval(np.tile(q0, (32, 1)), np.tile(t0, (32, 1)), 10, 100, kernel_pass=False)[0] for _ in range(3))
out['batch32_f64_step_us'] = ms / 100 * 1e3
out['c2_kernel_us'] = B.bench_kernel(q0, t0, 10, 300) * 1e3
print(json.dumps(out))
'''
from edge_alignment_amd import capi as _capi
key, val = sys.argv[1].split('=', 1)
libs = ['default', sys.argv[1]]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != 'default':
            env[key] = val
        o = subprocess.run([sys.executable, '-c', CHILD, _capi.LIB_PATH], capture_output=True, text=True, env=env)
        if o.returncode != 0:
            print(l, 'FAILED', o.stderr[-2000:]); sys.exit(1)
        res[l].append(json.loads(o.stdout.strip().splitlines()[-1]))
for l in libs:
    keys = res[l][0].keys()
    print(os.path.basename(l), {k: round(float(sorted(x[k] for x in res[l])[len(res[l]) // 2]), 2) for k in keys})

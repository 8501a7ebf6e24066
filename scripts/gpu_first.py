import sys, time, numpy as np
sys.path.insert(0, '.')
from edge_alignment_amd import capi, synth
from oracle import ea_oracle as eo
print('devices', capi.device_count(), capi.load().ea_version())
pr = synth.make_problem(120, 160, 5000, 40, 1, 130., 130., 79.5, 59.5,
                        planted_q=synth.quat_from_axis_angle([1,2,3], np.deg2rad(1.0)), planted_t=(0.01,-0.005,0.02), normalize=True)
K = pr['K']
O = eo.OracleProblem(pr['grid'], *K)
q0 = np.array([1.,0,0,0]); t0 = np.zeros(3)
oe = O.eval(pr['xyz'], q0, t0, materialize=True)
for dt, name, tol in ((capi.EA_F64,'f64',1e-11),(capi.EA_F32,'f32',2e-3)):
    P = capi.Problem(*K, dtype=dt)
    P.set_points(pr['xyz']); P.set_dt_grid(pr['grid'])
    for use_lds in (1, 0):
        b = capi.Batch([P]); b.set_tuning('use_lds', use_lds)
        ge = b.eval(q0, t0)
        relJ = np.abs(ge['JtJ'][0]-oe['JtJ']).max()/np.abs(oe['JtJ']).max()
        relg = np.abs(ge['Jtr'][0]-oe['Jtr']).max()/np.abs(oe['Jtr']).max()
        print(name, 'lds', use_lds, 'cost', ge['cost'][0], oe['cost'], 'relJtJ %.2e relJtr %.2e'%(relJ, relg), 'bad', ge['n_invalid'][0], oe['n_invalid'])
        b.close()
    r, J = P.eval_points(q0, t0, corrected=False)
    print(name, 'points r', np.nanmax(np.abs(r-oe['raw_r'])), 'J', np.nanmax(np.abs(J-oe['raw_J']))/np.nanmax(np.abs(oe['raw_J'])))
    q, t, s = P.solve(q0, t0)
    qo, to, so = O.solve(pr['xyz'], q0, t0)
    print(name, 'solve', s['why'], s['num_iterations'], s['final_cost'], '| oracle', so['why'], so['num_iterations'], so['final_cost'])
    print('   dq', synth.rotation_angle_between(q, qo), 'dt', np.linalg.norm(t-to), 'vs true', synth.rotation_angle_between(q, pr['q_true']), np.linalg.norm(t-pr['t_true']), 'ms', s['total_time_ms'])
    n = min(len(s['it_cost']), len(so['it_cost']))
    print('   trace max rel diff', np.max(np.abs(s['it_cost'][:n]-so['it_cost'][:n])/so['it_cost'][:n]))
    P.close()
# timing
for cfg, dt, name in ((synth.config_c2_twin(), capi.EA_F64, 'c2 f64'), (synth.config_c2_twin(), capi.EA_F32, 'c2 f32')):
    P = capi.Problem(*cfg['K'], dtype=dt); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid'])
    b = capi.Batch([P])
    for use_lds in (1,0):
        b.set_tuning('use_lds', use_lds)
        ms, msk = b.bench_eval(q0, t0, 20, 200)
        n = P.num_points
        print(name, 'lds', use_lds, 'tiles', b.info('num_tiles'), 'ppt', b.info('points_per_thread'), 'ms/step %.4f kernel %.4f  evals/s %.3e'%(ms/200, msk, n/(ms/200*1e-3)))
    t1=time.time(); q,t,s = P.solve(q0,t0); print('   solve', s['why'], s['num_iterations'], 'ms', s['total_time_ms'], 'it/s', s['num_iterations']/(s['total_time_ms']*1e-3))
    b.close(); P.close()
t1=time.time(); cfg = synth.config_c5(); print('c5 gen', time.time()-t1)
for dt, name in ((capi.EA_F32,'c5 f32'),(capi.EA_F64,'c5 f64')):
    P = capi.Problem(*cfg['K'], dtype=dt); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    b = capi.Batch([P])
    for use_lds in (1,0):
      for ppt in (1,2,4):
        b.set_tuning('use_lds', use_lds); b.set_tuning('points_per_thread', ppt)
        ms, msk = b.bench_eval(q0, t0, 10, 100)
        n = P.num_points
        print(name, 'lds', use_lds, 'ppt', ppt, 'tiles', b.info('num_tiles'), 'ms/step %.4f kernel %.4f  evals/s %.3e'%(ms/100, msk, n/(ms/100*1e-3)))
    q,t,s = P.solve(q0,t0); print('   solve', s['why'], s['num_iterations'], 'ms', s['total_time_ms'], 'err', synth.rotation_angle_between(q,cfg['q_true']), np.linalg.norm(t-cfg['t_true']))
    b.close(); P.close()

"""Phase shares of the fused kernel from in-kernel s_memtime stamps (diagnostic library only)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
from edge_alignment_amd import capi, synth
capi.LIB_PATH = capi.LIB_PATH.replace('libea_hip.so', 'libea_hip_stamps.so')
L = capi.load()
L.ea_debug_eval_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
q0 = np.array([1., 0, 0, 0]); t0 = np.zeros(3)
def run(name, cfg, dtype, loss, tune=None):
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(*loss)
    B = capi.Batch([P])
    for k, v in (tune or {}).items(): B.set_tuning(k, v)
    cap = 1 << 16
    st = np.zeros((cap, 8), dtype=np.uint64); n = C.c_int64(); g = (C.c_int * 2)()
    rc = L.ea_debug_eval_stamps(B._h, q0.ctypes.data_as(C.POINTER(C.c_double)), t0.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n), g)
    assert rc == 0, L.ea_last_error()
    st = st[:n.value].astype(np.int64)
    ok = st[:, 7] > 0
    st = st[ok]
    t_first = st[:, 0].min()
    d = np.diff(st, axis=1)
    names = ['desc+pose', 'points', 'project', 'sample+acc', 'wave reduce', 'barrier', 'fold+store']
    print('%s: %d workgroups (%d with work); cycles (100 MHz? see ratio) median per phase:' % (name, n.value, ok.sum()))
    for i, nm in enumerate(names):
        print('   %-12s median %7.0f  p90 %7.0f' % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
    print('   workgroup total median %.0f; first start -> last end %.0f; start spread %.0f' % (np.median(st[:, 7] - st[:, 0]), st[:, 7].max() - t_first, st[:, 0].max() - t_first))
    B.close(); P.close()
if len(sys.argv) > 1 and sys.argv[1] == 'eval':   # (needs a library built with -DEA_STAMPS -DEA_STAMPS_EVAL)
    run('c2 f64', synth.config_c2_twin(), capi.EA_F64, (capi.LOSS_CAUCHY, 1.0))
    run('lm1e5 f64', synth.config_c2_twin(seed=7, n_points=100000), capi.EA_F64, (capi.LOSS_CAUCHY, 1.0))
    run('c5 f32', synth.config_c5(), capi.EA_F32, (capi.LOSS_TRIVIAL, 1.0))

def run_lm(name, cfg):
    P = capi.Problem(*cfg['K'], dtype=capi.EA_F64); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    B = capi.Batch([P])
    B.set_tuning('fused_iterations', 0)   # the (evaluate, step) pairs
    q, t, summ = B.solve(q0, t0)   # warm
    assert L.ea_debug_lm_stamps_begin() == 0
    q, t, summ = B.solve(q0, t0)
    st = np.zeros((128, 8), dtype=np.uint64)
    assert L.ea_debug_lm_stamps_end(st.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    st = st.astype(np.int64); pr = st[64:]; st = st[:64]
    st_all = st
    keep = (st[:, 5] > 0) & (pr[:, 5] > 0) & (pr[:, 1] > 0)
    pr = pr[keep]; st = st[keep]
    seq = np.concatenate([st[:, 2:3], pr[:, :2], pr[:, 6:8], pr[:, 2:6], st[:, 3:4]], axis=1)
    dd = np.diff(seq, axis=1)
    if len(dd) == 0:
        # (the usual iteration runs lm_advance_fast, which carries no probes: only launches that took the general form have them)
        print('      no iteration with state-machine probes captured (the usual iteration takes lm_advance_fast)')
        st = st_all[st_all[:, 5] > 0]
    for i, nm in enumerate([] if len(dd) == 0 else ['tests+rel', 'take_system', 'radius update', 'trace', 'checks+scale', 'strategy step', 'model change', 'pose_plus', 'tail']):
        print('      probe %-14s %7.0f %7.0f' % (nm, np.median(dd[:, i]), dd[:, i].max()))
    d = np.diff(st[:, :6], axis=1)
    names = ['running load', 'stage+fold', 'lm_advance', 'make_pose_state', 'publish+store']
    print('%s LM step kernel (%d iterations captured), cycles median / max:' % (name, len(st)))
    for i, nm in enumerate(names):
        print('   %-16s %7.0f %7.0f' % (nm, np.median(d[:, i]), d[:, i].max()))
    print('   total median %.0f; eval-to-eval period median %.0f' % (np.median(st[:, 5] - st[:, 0]), np.median(np.diff(np.sort(st[:, 0])))))
    P.close()
run_lm('lm1e5 f64', synth.config_c2_twin(seed=7, n_points=100000))


def run_lm_fused(name, cfg, dtype=capi.EA_F64):
    """ea_lm_iter_kernel (one launch per iteration): stamps of workgroup 0 (an evaluator), lane 0"""
    P = capi.Problem(*cfg['K'], dtype=dtype); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); P.set_loss(capi.LOSS_CAUCHY, 1.0)
    B = capi.Batch([P])
    q, t, summ = B.solve(q0, t0)   # warm
    assert B.info('fused_iterations') == 1
    assert L.ea_debug_lm_stamps_begin() == 0
    q, t, summ = B.solve(q0, t0)
    st = np.zeros((128, 8), dtype=np.uint64)
    assert L.ea_debug_lm_stamps_end(st.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    st = st.astype(np.int64)[:64]
    st = st[(st[:, 5] > 0) & (st[:, 0] > 0)]
    st = st[np.argsort(st[:, 0])]
    d = np.diff(st[:, :6], axis=1)
    print('%s fused LM iteration kernel (%d launches captured), s_memtime ticks median / max:' % (name, len(st)))
    for i, nm in enumerate(['entry -> folded', 'state machine', 'pose -> LDS + barrier', 'evaluate', 'row store']):
        print('   %-22s %7.0f %7.0f' % (nm, np.median(d[:, i]), d[:, i].max()))
    print('   in-kernel total median %.0f; launch-to-launch period median %.0f; gap (row stored -> entry of the next) median %.0f'
          % (np.median(st[:, 5] - st[:, 0]), np.median(np.diff(st[:, 0])), np.median(st[1:, 0] - st[:-1, 5])))
    P.close()


run_lm_fused('c1-size 1482 f64', synth.config_c2_twin(seed=7, n_points=1482))
run_lm_fused('c2 5e4 f64', synth.config_c2_twin())
run_lm_fused('lm1e5 f64', synth.config_c2_twin(seed=7, n_points=100000))

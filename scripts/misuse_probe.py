"""C-ABI misuse probe: every entry point with a valid handle and one bad argument (NULL where data is expected, negative
or zero extents, unknown enum values, NaN / zero poses, stale state).  A call must return an error code (or a documented
result) -- never crash, hang or corrupt the handle: after each bad call the same handle must still evaluate a known
problem to the same bits.  Each case runs in a forked child (after the parent has NOT touched the GPU), so a crash is a
line in the report.  usage: python scripts/misuse_probe.py [case-substring]"""
import ctypes as C, os, sys, signal, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sel = sys.argv[1] if len(sys.argv) > 1 else ""


def cases():
    """-> list of (name, fn(ctx) -> rc or None).  ctx: lib, P (valid problem handle with points + DT), B (batch), arrays"""
    L = []
    def case(name):
        def deco(f):
            L.append((name, f)); return f
        return deco
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    NULLD = C.POINTER(C.c_double)()

    @case("create: null out")
    def _(c): return c.lib.ea_problem_create(None, C.byref(c.cam), 0, 0)
    @case("create: dtype 7")
    def _(c):
        h = C.c_void_p(); return c.lib.ea_problem_create(C.byref(h), C.byref(c.cam), 7, 0)
    @case("create: device 99")
    def _(c):
        h = C.c_void_p(); return c.lib.ea_problem_create(C.byref(h), C.byref(c.cam), 0, 99)
    @case("create: device -1")
    def _(c):
        h = C.c_void_p(); return c.lib.ea_problem_create(C.byref(h), C.byref(c.cam), 0, -1)
    @case("create: fx = 0 / NaN camera")
    def _(c):
        h = C.c_void_p(); cam = type(c.cam)(0.0, float("nan"), 1.0, 1.0); return c.lib.ea_problem_create(C.byref(h), C.byref(cam), 0, 0)
    @case("set_points: null with n > 0")
    def _(c): return c.lib.ea_problem_set_points(c.P, NULLD, 100, 3)
    @case("set_points: n < 0")
    def _(c): return c.lib.ea_problem_set_points(c.P, dp(c.xyz), -5, 3)
    @case("set_points: stride 2")
    def _(c): return c.lib.ea_problem_set_points(c.P, dp(c.xyz), 10, 2)
    @case("set_points: stride 0 / negative")
    def _(c): return c.lib.ea_problem_set_points(c.P, dp(c.xyz), 10, 0) or c.lib.ea_problem_set_points(c.P, dp(c.xyz), 10, -3)
    @case("set_points: NaN / Inf coordinates (accepted or refused, then eval must not crash)")
    def _(c):
        X = c.xyz.copy(); X[5] = [np.nan, 1, 2]; X[9] = [np.inf, -np.inf, 1]; X[11] = [0, 0, 0]
        rc = c.lib.ea_problem_set_points(c.P, dp(X), len(X), 3)
        cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
        rc2 = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad))
        c.note = "set rc %d eval rc %d cost %r invalid %d" % (rc, rc2, cost.value, bad.value)
        c.lib.ea_problem_set_points(c.P, dp(c.xyz), len(c.xyz), 3)
        return -1 if (rc or rc2 == 0) else 0
    @case("set_points: n = 0 then eval")
    def _(c):
        rc = c.lib.ea_problem_set_points(c.P, NULLD, 0, 3)
        cost = C.c_double(-7); bad = C.c_int64(); J = np.ones(36); g = np.ones(6)
        rc2 = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad))
        c.note = "set rc %d eval rc %d cost %r" % (rc, rc2, cost.value)
        c.lib.ea_problem_set_points(c.P, dp(c.xyz), len(c.xyz), 3)
        return -1 if (rc == 0 and rc2 == 0 and cost.value == 0.0 and not J.any()) else 0
    @case("set_points_device: null pointers")
    def _(c): return c.lib.ea_problem_set_points_device(c.P, None, None, None, 100)
    @case("set_points_device: n < 0")
    def _(c): return c.lib.ea_problem_set_points_device(c.P, None, None, None, -1)
    @case("set_dt: null")
    def _(c): return c.lib.ea_problem_set_dt(c.P, NULLD, 10, 10)
    @case("set_dt: rows 0 / negative")
    def _(c): return c.lib.ea_problem_set_dt(c.P, dp(c.grid), 0, 10) or c.lib.ea_problem_set_dt(c.P, dp(c.grid), -4, 10) or c.lib.ea_problem_set_dt(c.P, dp(c.grid), 10, -1)
    @case("set_dt: overflowing extents")
    def _(c): return c.lib.ea_problem_set_dt(c.P, dp(c.grid), 2 ** 31 - 1, 2 ** 31 - 1)
    @case("set_dt: 1x1 grid (then eval)")
    def _(c):
        g1 = np.array([[0.25]])
        rc = c.lib.ea_problem_set_dt(c.P, dp(g1), 1, 1)
        cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
        rc2 = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad))
        c.note = "set rc %d eval rc %d cost %r" % (rc, rc2, cost.value)
        c.lib.ea_problem_set_dt(c.P, dp(c.grid), *c.grid.shape)
        return -1 if rc != 0 or (rc2 == 0 and not J.any()) else 0
    @case("set_dt: NaN texels (eval must not crash)")
    def _(c):
        g2 = c.grid.copy(); g2[3:6, 3:6] = np.nan
        rc = c.lib.ea_problem_set_dt(c.P, dp(g2), *g2.shape)
        cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
        rc2 = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad))
        o = c.opts(); s = c.Summary(); q = c.q.copy(); t = c.t.copy()
        rc3 = c.lib.ea_solve(c.P, C.byref(o), dp(q), dp(t), C.byref(s))
        c.note = "eval rc %d cost %r; solve rc %d termination %d iterations %d" % (rc2, cost.value, rc3, s.termination, s.num_iterations)
        c.lib.ea_problem_set_dt(c.P, dp(c.grid), *c.grid.shape)
        return -1
    @case("set_dt_image_device: null / bad extents")
    def _(c): return c.lib.ea_problem_set_dt_image_device(c.P, None, 10, 10) or c.lib.ea_problem_set_dt_image_device(c.P, None, 0, 0)
    @case("set_loss: kind -1 / 3, a <= 0, a NaN")
    def _(c):
        rcs = [c.lib.ea_problem_set_loss(c.P, -1, 1.0), c.lib.ea_problem_set_loss(c.P, 3, 1.0), c.lib.ea_problem_set_loss(c.P, 1, -1.0),
               c.lib.ea_problem_set_loss(c.P, 2, float("nan"))]
        c.note = str(rcs); return -1 if all(r != 0 for r in rcs) else 0
    @case("set_flavour: negative / NaN guard")
    def _(c):
        rcs = [c.lib.ea_problem_set_flavour(c.P, -1.0, 0.0, 0), c.lib.ea_problem_set_flavour(c.P, float("nan"), 0.0, 0), c.lib.ea_problem_set_flavour(c.P, 0.01, float("inf"), 0)]
        c.lib.ea_problem_set_flavour(c.P, 0.01, 0.0, 0)
        c.note = str(rcs); return -1 if all(r != 0 for r in rcs) else 0
    @case("set_distortion: NaN")
    def _(c):
        rc = c.lib.ea_problem_set_distortion(c.P, float("nan"), 0, 0, 0, 0)
        c.lib.ea_problem_set_distortion(c.P, 0, 0, 0, 0, 0); return rc
    @case("set_second_camera: null = back to the first camera (documented); non-affine / NaN matrices refused")
    def _(c):
        r0 = c.lib.ea_problem_set_second_camera(c.P, NULLD, NULLD)
        Z = np.zeros(16); r1 = c.lib.ea_problem_set_second_camera(c.P, dp(Z), dp(Z))
        N = np.eye(4).reshape(16); N[3] = np.nan; r2 = c.lib.ea_problem_set_second_camera(c.P, dp(N), dp(np.eye(4).reshape(16)))
        c.note = "null %d zero matrix %d NaN %d" % (r0, r1, r2)
        return -1 if r0 == 0 and r1 != 0 and r2 != 0 else 0
    @case("add_term: null / self / twice / cycle")
    def _(c):
        rcs = [c.lib.ea_problem_add_term(c.P, None), c.lib.ea_problem_add_term(c.P, c.P)]
        r1 = c.lib.ea_problem_add_term(c.P, c.P2); r2 = c.lib.ea_problem_add_term(c.P, c.P2); r3 = c.lib.ea_problem_add_term(c.P2, c.P)
        c.lib.ea_problem_clear_terms(c.P); c.lib.ea_problem_clear_terms(c.P2)
        c.note = "null/self %s first %d twice %d cycle %d" % (rcs, r1, r2, r3)
        return -1 if all(r != 0 for r in rcs) and r1 == 0 and r2 != 0 and r3 != 0 else 0   # twice: refused ("term already added")
    @case("add_term: term destroyed before eval")
    def _(c):
        h = C.c_void_p(); c.lib.ea_problem_create(C.byref(h), C.byref(c.cam), 0, 0)
        c.lib.ea_problem_set_points(h, dp(c.xyz), 50, 3); c.lib.ea_problem_set_dt(h, dp(c.grid), *c.grid.shape)
        r1 = c.lib.ea_problem_add_term(c.P, h)
        c.lib.ea_problem_destroy(h)
        cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
        rc2 = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad))
        c.note = "add rc %d, eval after the term's destroy rc %d cost %r" % (r1, rc2, cost.value)
        c.lib.ea_problem_clear_terms(c.P)
        return -1
    @case("eval: null q / t / outputs")
    def _(c):
        cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
        rcs = [c.lib.ea_eval(c.P, NULLD, dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad)),
               c.lib.ea_eval(c.P, dp(c.q), NULLD, C.byref(cost), dp(J), dp(g), C.byref(bad))]
        r_nullable = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), None, NULLD, NULLD, None)
        c.note = "null q/t %s; all outputs null rc %d" % (rcs, r_nullable)
        return -1 if all(r != 0 for r in rcs) else 0
    @case("eval: zero / NaN / Inf quaternion, NaN t")
    def _(c):
        out = []
        for q, t in ((np.zeros(4), c.t), (np.array([np.nan, 0, 0, 0]), c.t), (np.array([np.inf, 0, 0, 0]), c.t), (c.q, np.array([np.nan, 0, 0])), (c.q * 1e200, c.t)):
            cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
            rc = c.lib.ea_eval(c.P, dp(np.ascontiguousarray(q, dtype=np.float64)), dp(np.ascontiguousarray(t, dtype=np.float64)), C.byref(cost), dp(J), dp(g), C.byref(bad))
            out.append((rc, cost.value, bad.value))
        c.note = str(out); return -1
    @case("eval_points: null outputs")
    def _(c): return c.lib.ea_eval_points(c.P, dp(c.q), dp(c.t), NULLD, NULLD, 1) or -1
    @case("cost: null")
    def _(c): return c.lib.ea_cost(c.P, NULLD, NULLD, None, None)
    @case("pixel_cost: null out")
    def _(c): return c.lib.ea_problem_pixel_cost(c.P, dp(c.q), dp(c.t), None)
    @case("solve: null q / null opt (defaults) / null summary")
    def _(c):
        o = c.opts(); s = c.Summary(); q = c.q.copy(); t = c.t.copy()
        r1 = c.lib.ea_solve(c.P, C.byref(o), NULLD, dp(t), C.byref(s))
        r2 = c.lib.ea_solve(c.P, None, dp(q), dp(t), C.byref(s))
        q = c.q.copy(); t = c.t.copy()
        r3 = c.lib.ea_solve(c.P, C.byref(o), dp(q), dp(t), None)
        c.note = "null q %d, null opt %d (iterations %d), null summary %d" % (r1, r2, s.num_iterations, r3)
        return -1 if r1 != 0 else 0
    @case("solve: options out of range")
    def _(c):
        out = []
        for k, v in (("max_num_iterations", -1), ("max_num_iterations", 0), ("max_num_iterations", 10 ** 6), ("strategy", 9), ("initial_trust_region_radius", -1.0),
                     ("initial_trust_region_radius", float("nan")), ("function_tolerance", float("nan")), ("function_tolerance", -1.0),
                     ("min_relative_decrease", 2.0), ("solve_timeout_ms", float("nan"))):
            o = c.opts()
            if not hasattr(o, k):
                out.append((k, "absent")); continue
            setattr(o, k, v)
            s = c.Summary(); q = c.q.copy(); t = c.t.copy()
            rc = c.lib.ea_solve(c.P, C.byref(o), dp(q), dp(t), C.byref(s))
            out.append((k, v, rc, s.termination, s.num_iterations))
        c.note = str(out); return -1
    @case("solve: zero / NaN start pose")
    def _(c):
        out = []
        for q0 in (np.zeros(4), np.array([np.nan, 0, 0, 0.0]), np.array([1e-300, 0, 0, 0])):
            o = c.opts(); s = c.Summary(); q = q0.copy(); t = c.t.copy()
            rc = c.lib.ea_solve(c.P, C.byref(o), dp(q), dp(t), C.byref(s))
            out.append((rc, s.termination, s.num_iterations, q.tolist()))
        c.note = str(out); return -1
    @case("solve_pyramid: null level inside, nlevels 0 / negative / huge")
    def _(c):
        arr = (C.c_void_p * 3)(c.P, None, c.P2)
        s = c.Summary(); q = c.q.copy(); t = c.t.copy()
        rcs = [c.lib.ea_solve_pyramid(arr, 3, None, dp(q), dp(t), C.byref(s)), c.lib.ea_solve_pyramid(arr, 0, None, dp(q), dp(t), C.byref(s)),
               c.lib.ea_solve_pyramid(arr, -2, None, dp(q), dp(t), C.byref(s))]
        c.note = str(rcs); return -1 if all(r != 0 for r in rcs) else 0
    @case("batch_create: null entry / count 0 / negative / duplicate handles")
    def _(c):
        h = C.c_void_p()
        arr = (C.c_void_p * 2)(c.P, None)
        rcs = [c.lib.ea_batch_create(C.byref(h), arr, 2), c.lib.ea_batch_create(C.byref(h), arr, 0), c.lib.ea_batch_create(C.byref(h), arr, -1)]
        dup = (C.c_void_p * 2)(c.P, c.P)
        rd = c.lib.ea_batch_create(C.byref(h), dup, 2)
        note = "null/0/-1 %s duplicate rc %d" % (rcs, rd)
        if rd == 0:
            cost = np.zeros(2); J = np.zeros(72); g = np.zeros(12); bad = np.zeros(2, np.int64)
            q2 = np.stack([c.q, c.q]); t2 = np.stack([c.t, c.t])
            re = c.lib.ea_batch_eval(h, dp(q2), dp(t2), dp(cost), dp(J), dp(g), bad.ctypes.data_as(C.POINTER(C.c_int64)))
            note += " eval rc %d costs %s" % (re, cost.tolist()); c.lib.ea_batch_destroy(h)
        c.note = note
        return -1 if all(r != 0 for r in rcs) else 0
    @case("batch: problem destroyed while in a batch, then batch eval")
    def _(c):
        h = C.c_void_p(); c.lib.ea_problem_create(C.byref(h), C.byref(c.cam), 0, 0)
        c.lib.ea_problem_set_points(h, dp(c.xyz), 50, 3); c.lib.ea_problem_set_dt(h, dp(c.grid), *c.grid.shape)
        b = C.c_void_p(); arr = (C.c_void_p * 2)(c.P, h)
        r0 = c.lib.ea_batch_create(C.byref(b), arr, 2)
        c.lib.ea_problem_destroy(h)
        cost = np.zeros(2); J = np.zeros(72); g = np.zeros(12); bad = np.zeros(2, np.int64)
        q2 = np.stack([c.q, c.q]); t2 = np.stack([c.t, c.t])
        re = c.lib.ea_batch_eval(b, dp(q2), dp(t2), dp(cost), dp(J), dp(g), bad.ctypes.data_as(C.POINTER(C.c_int64)))
        c.note = "create %d eval after a member's destroy rc %d" % (r0, re)
        c.lib.ea_batch_destroy(b)
        return -1
    @case("batch_eval / batch_solve: null arrays")
    def _(c):
        rcs = [c.lib.ea_batch_eval(c.B, NULLD, NULLD, NULLD, NULLD, NULLD, None), c.lib.ea_batch_solve(c.B, None, NULLD, NULLD, None)]
        c.note = str(rcs); return -1 if all(r != 0 for r in rcs) else 0
    @case("batch tuning: unknown key / null key / absurd values")
    def _(c):
        rcs = [c.lib.ea_batch_set_tuning(c.B, b"no_such_key", 1), c.lib.ea_batch_set_tuning(c.B, None, 1)]
        vals = []
        for k in (b"solve_streams", b"points_per_thread", b"lds_texels", b"threads", b"buffer_loads", b"chunk"):
            for v in (-1, 0, 3, 10 ** 9):
                r = c.lib.ea_batch_set_tuning(c.B, k, v)
                cost = np.zeros(1); J = np.zeros(36); g = np.zeros(6); bad = np.zeros(1, np.int64)
                re = c.lib.ea_batch_eval(c.B, dp(c.q), dp(c.t), dp(cost), dp(J), dp(g), bad.ctypes.data_as(C.POINTER(C.c_int64)))
                vals.append((k.decode(), v, r, re, cost[0] == c.ref_cost or re != 0))
        c.note = "unknown/null %s; " % rcs + str([v for v in vals if not v[4]] or "every accepted value evaluates to the reference bits")
        v = C.c_int64(); r = c.lib.ea_batch_get_info(c.B, b"nope", C.byref(v)); r2 = c.lib.ea_batch_get_info(c.B, None, C.byref(v)); r3 = c.lib.ea_batch_get_info(c.B, b"num_tiles", None)
        c.note += " info unknown/null/nullout %s" % [r, r2, r3]
        return -1 if all(x != 0 for x in rcs) and all(x[4] for x in vals) else 0
    @case("bench hooks: steps 0 / negative, steps before capture")
    def _(c):
        ms = C.c_double()
        rcs = [c.lib.ea_batch_bench_steps(c.B, 0, None), c.lib.ea_batch_bench_steps(c.B, -3, None), c.lib.ea_batch_bench_capture(c.B, 0), c.lib.ea_batch_bench_capture(c.B, -1),
               c.lib.ea_batch_bench_fold(c.B, -1, 0, C.byref(ms)), c.lib.ea_batch_bench_kernel(c.B, dp(c.q), dp(c.t), 0, 0, C.byref(ms))]
        c.note = str(rcs); return -1
    @case("frames: null images / tiny / negative extents")
    def _(c):
        u8 = C.POINTER(C.c_uint8); u16 = C.POINTER(C.c_uint16)
        img = np.zeros((8, 8, 3), np.uint8); d = np.zeros((8, 8), np.uint16)
        ip = img.ctypes.data_as(u8); dq = d.ctypes.data_as(u16)
        rcs = [c.lib.ea_problem_set_ref_frame(c.P, None, dq, 8, 8, 5000.0, 35), c.lib.ea_problem_set_ref_frame(c.P, ip, None, 8, 8, 5000.0, 35),
               c.lib.ea_problem_set_ref_frame(c.P, ip, dq, -8, 8, 5000.0, 35), c.lib.ea_problem_set_ref_frame(c.P, ip, dq, 8, 0, 5000.0, 35),
               c.lib.ea_problem_set_ref_frame(c.P, ip, dq, 2, 2, 5000.0, 35), c.lib.ea_problem_set_ref_frame(c.P, ip, dq, 8, 8, 0.0, 35),
               c.lib.ea_problem_set_ref_frame(c.P, ip, dq, 8, 8, float("nan"), 35), c.lib.ea_problem_set_ref_frame(c.P, ip, dq, 60000, 60000, 5000.0, 35),
               c.lib.ea_problem_set_now_frame(c.P, None, 8, 8, 35, 1, 1), c.lib.ea_problem_set_now_frame(c.P, ip, 8, -1, 35, 1, 1),
               c.lib.ea_problem_set_now_frame_canny(c.P, None, None, 8, 8, 30.0, 90.0, 1, 0.0, 1.0),
               c.lib.ea_problem_set_now_frame_canny(c.P, ip, None, 8, 8, float("nan"), 90.0, 1, 0.0, 1.0),
               c.lib.ea_problem_set_now_frame_ros(c.P, None, 8, 8, 150.0, 100.0),
               c.lib.ea_problem_set_ref_frame_ros_scaled(c.P, ip, None, 8, 8, 1, 150.0, 100.0),
               c.lib.ea_problem_set_now_frame_ros_scaled(c.P, ip, 8, 8, -1, 150.0, 100.0), c.lib.ea_problem_set_now_frame_ros_scaled(c.P, ip, 8, 8, 40, 150.0, 100.0),
               c.lib.ea_resize_half(0, 5, ip, 8, 8, ip), c.lib.ea_resize_half(0, 0, None, 8, 8, ip), c.lib.ea_resize_half(0, 0, ip, 7, 8, ip), c.lib.ea_resize_half(42, 0, ip, 8, 8, ip)]
        c.note = str(rcs)
        c.restore()
        return -1 if all(r != 0 for r in rcs) else 0
    @case("get_points / get_dt: small capacity, null")
    def _(c):
        out = np.zeros(6); h = C.c_int(); w = C.c_int()
        rcs = [c.lib.ea_problem_get_points(c.P, dp(out), 2), c.lib.ea_problem_get_points(c.P, NULLD, 1000), c.lib.ea_problem_get_points(c.P, dp(out), -1),
               c.lib.ea_problem_get_dt(c.P, NULLD, None, None)]
        r_size = c.lib.ea_problem_get_dt(c.P, NULLD, C.byref(h), C.byref(w))
        c.note = "%s; size query rc %d -> %dx%d" % (rcs, r_size, h.value, w.value); return -1 if all(r != 0 for r in rcs[:3]) else 0
    @case("tracker: bad frames on a live tracker")
    def _(c):
        tr = C.c_void_p(); r0 = c.lib.ea_tracker_create(C.byref(tr), C.byref(c.cam), 0, 0, 0)
        u8 = C.POINTER(C.c_uint8); u16 = C.POINTER(C.c_uint16)
        q = c.q.copy(); t = c.t.copy(); s = c.Summary(); solved = C.c_int()
        rcs = [c.lib.ea_tracker_push_frame(tr, None, None, 48, 64, 5000.0, None, dp(q), dp(t), C.byref(s), C.byref(solved)),
               c.lib.ea_tracker_push_frame(tr, np.zeros((48, 64, 3), np.uint8).ctypes.data_as(u8), np.zeros((48, 64), np.uint16).ctypes.data_as(u16), 48, 64, 5000.0, None, NULLD, dp(t), C.byref(s), C.byref(solved)),
               c.lib.ea_tracker_push_frame(tr, np.zeros((48, 64, 3), np.uint8).ctypes.data_as(u8), np.zeros((48, 64), np.uint16).ctypes.data_as(u16), 0, 64, 5000.0, None, dp(q), dp(t), C.byref(s), C.byref(solved))]
        # a featureless first frame, then a featureless second one: no points, no edges
        r_flat = [c.lib.ea_tracker_push_frame(tr, np.full((48, 64, 3), 9, np.uint8).ctypes.data_as(u8), np.full((48, 64), 1000, np.uint16).ctypes.data_as(u16), 48, 64, 5000.0, None, dp(q), dp(t), C.byref(s), C.byref(solved)) for _ in range(2)]
        c.note = "create %d bad pushes %s flat pushes %s solved %d" % (r0, rcs, r_flat, solved.value)
        c.lib.ea_tracker_destroy(tr)
        return -1 if all(r != 0 for r in rcs) else 0
    @case("destroy: null handles, double clear")
    def _(c):
        c.lib.ea_problem_destroy(None); c.lib.ea_batch_destroy(None); c.lib.ea_tracker_destroy(None)
        return c.lib.ea_problem_clear_terms(None) or c.lib.ea_problem_clear_terms(c.P) or -1
    @case("selftest / device_count: null")
    def _(c): return c.lib.ea_device_count(None) or c.lib.ea_selftest_wave_reduce(0, None, None, None, None)
    @case("sharded: null callback, callback that does nothing on a valid handle")
    def _(c):
        from edge_alignment_amd import capi
        s = c.Summary(); q = c.q.copy(); t = c.t.copy()
        r1 = c.lib.ea_solve_sharded(c.P, None, C.cast(None, capi.ALLREDUCE_FN), None, dp(q), dp(t), C.byref(s))
        r2 = c.lib.ea_solve_sharded_device(c.P, None, C.cast(None, capi.DEVICE_ALLREDUCE_FN), None, None, dp(q), dp(t), C.byref(s)) if hasattr(capi, "DEVICE_ALLREDUCE_FN") else "n/a"
        c.note = "%s %s" % (r1, r2); return -1 if r1 != 0 else 0
    # ---- materialised mode and the pipelined measurement hook (round 2, second session)
    VP = C.c_void_p
    I64 = C.POINTER(C.c_int64)
    @case("eval_rows: null outputs / null pose / bad layout / capacity too small")
    def _(c):
        n = C.c_int64(); bad = C.c_int64()
        assert c.lib.ea_problem_num_rows(c.P, C.byref(n)) == 0 and n.value == len(c.xyz)
        r = np.zeros(n.value); J = np.zeros((n.value, 6))
        rp, Jp = r.ctypes.data_as(VP), J.ctypes.data_as(VP)
        rcs = [c.lib.ea_eval_rows(c.P, dp(c.q), dp(c.t), 1, 0, None, None, n.value, C.byref(bad)),
               c.lib.ea_eval_rows(c.P, NULLD, dp(c.t), 1, 0, rp, Jp, n.value, C.byref(bad)),
               c.lib.ea_eval_rows(c.P, dp(c.q), dp(c.t), 1, 2, rp, Jp, n.value, C.byref(bad)),
               c.lib.ea_eval_rows(c.P, dp(c.q), dp(c.t), 1, -1, rp, Jp, n.value, C.byref(bad)),
               c.lib.ea_eval_rows(c.P, dp(c.q), dp(c.t), 1, 0, rp, Jp, n.value - 1, C.byref(bad)),
               c.lib.ea_eval_rows(None, dp(c.q), dp(c.t), 1, 0, rp, Jp, n.value, C.byref(bad)),
               c.lib.ea_problem_num_rows(c.P, None), c.lib.ea_problem_num_rows(None, C.byref(n))]
        c.note = str(rcs)
        ok = c.lib.ea_eval_rows(c.P, dp(c.q), dp(c.t), 1, 0, rp, Jp, n.value, None)   # n_invalid may be NULL
        return -1 if (all(x != 0 for x in rcs) and ok == 0 and np.isfinite(J).all() and J.any()) else 0
    @case("eval_rows_device: one pointer null / misaligned / host memory where the runtime can tell / capacity")
    def _(c):
        import torch
        n = len(c.xyz); bad = C.c_int64()
        r = torch.zeros(n + 2, dtype=torch.float64, device="cuda"); J = torch.zeros((n + 2, 6), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rp, Jp = r.data_ptr(), J.data_ptr()
        rcs = [c.lib.ea_eval_rows_device(c.P, dp(c.q), dp(c.t), 1, 0, VP(rp), None, n, C.byref(bad)),
               c.lib.ea_eval_rows_device(c.P, dp(c.q), dp(c.t), 1, 0, None, VP(Jp), n, C.byref(bad)),
               c.lib.ea_eval_rows_device(c.P, dp(c.q), dp(c.t), 1, 0, VP(rp + 8), VP(Jp), n, C.byref(bad)),
               c.lib.ea_eval_rows_device(c.P, dp(c.q), dp(c.t), 1, 0, VP(rp), VP(Jp), n - 1, C.byref(bad)),
               c.lib.ea_eval_rows_device(c.P, dp(c.q), dp(c.t), 1, 3, VP(rp), VP(Jp), n, C.byref(bad))]
        c.note = str(rcs)
        ok = c.lib.ea_eval_rows_device(c.P, dp(c.q), dp(c.t), 1, 0, VP(rp), VP(Jp), n + 2, C.byref(bad))
        tail_untouched = float(r[n:].abs().max()) == 0.0 and float(J[n:].abs().max()) == 0.0
        return -1 if (all(x != 0 for x in rcs) and ok == 0 and tail_untouched and float(J[:n].abs().max()) > 0) else 0
    @case("batch rows: null offsets / null batch; bench hooks: steps 0, no poses, null result arrays")
    def _(c):
        off = np.zeros(2, dtype=np.int64)
        b2 = C.c_void_p(); arr = (C.c_void_p * 1)(c.P2); assert c.lib.ea_batch_create(C.byref(b2), arr, 1) == 0
        ms = C.c_double()
        rcs = [c.lib.ea_batch_row_offsets(c.B, None), c.lib.ea_batch_row_offsets(None, off.ctypes.data_as(I64)),
               c.lib.ea_batch_bench_capture_pipelined(c.B, 0), c.lib.ea_batch_bench_capture_pipelined(None, 4),
               c.lib.ea_batch_bench_capture_pipelined(b2, 4),      # no poses uploaded on this batch yet
               c.lib.ea_batch_bench_steps_riding(c.B, 0, None), c.lib.ea_batch_bench_steps_riding(None, 4, None),
               c.lib.ea_batch_bench_steps_riding(b2, 4, None),     # no poses uploaded on this batch yet
               c.lib.ea_batch_bench_result(None, None, None, None, None), c.lib.ea_batch_bench_result(b2, None, None, None, None),
               c.lib.ea_batch_bench_result_riding(c.B, None, None, None, None),   # nothing pipelined captured
               c.lib.ea_batch_bench_rows(c.B, dp(c.q), dp(c.t), 1, 0, 1, None, None, 0, 0, 0, C.byref(ms)),   # launches 0
               c.lib.ea_batch_bench_rows(c.B, dp(c.q), dp(c.t), 1, 0, 1, None, None, 0, 0, 3, None)]
        c.note = str(rcs)
        c.lib.ea_batch_destroy(b2)
        return -1 if all(x != 0 for x in rcs) else 0
    return L


class Ctx:
    pass


def make_ctx():
    import torch
    torch.cuda.init()
    from edge_alignment_amd import capi, synth
    c = Ctx()
    c.lib = capi.load()
    c.cam = capi.Camera(120.0, 121.0, 63.5, 47.5)
    pr = synth.make_problem(96, 128, 2000, 12, 5, 120.0, 121.0, 63.5, 47.5, planted_q=synth.quat_from_axis_angle([1, 2, 3], 0.01), planted_t=(0.01, 0, 0.005), normalize=True)
    c.xyz = np.ascontiguousarray(pr["xyz"]); c.grid = np.ascontiguousarray(pr["grid"])
    c.q = np.array([1.0, 0, 0, 0]); c.t = np.zeros(3)
    c.Summary = capi.Summary
    c.opts = lambda: capi.default_options()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    def mk(n):
        h = C.c_void_p(); assert c.lib.ea_problem_create(C.byref(h), C.byref(c.cam), 0, 0) == 0
        assert c.lib.ea_problem_set_points(h, dp(c.xyz), n, 3) == 0 and c.lib.ea_problem_set_dt(h, dp(c.grid), *c.grid.shape) == 0
        return h
    c.P, c.P2 = mk(len(c.xyz)), mk(500)
    def restore():
        assert c.lib.ea_problem_set_points(c.P, dp(c.xyz), len(c.xyz), 3) == 0 and c.lib.ea_problem_set_dt(c.P, dp(c.grid), *c.grid.shape) == 0
        c.lib.ea_problem_set_loss(c.P, 1, 1.0)
    c.restore = restore
    b = C.c_void_p(); arr = (C.c_void_p * 1)(c.P); assert c.lib.ea_batch_create(C.byref(b), arr, 1) == 0
    c.B = b
    def ref():
        cost = C.c_double(); bad = C.c_int64(); J = np.zeros(36); g = np.zeros(6)
        rc = c.lib.ea_eval(c.P, dp(c.q), dp(c.t), C.byref(cost), dp(J), dp(g), C.byref(bad))
        return rc, cost.value, J.copy()
    c.ref = ref
    rc, c.ref_cost, c.ref_J = ref()
    assert rc == 0 and c.ref_cost > 0
    c.note = ""
    return c


def run_case(c, fn):
    """-> (verdict, message): "ok" = refused (or behaved as documented) and the handle still evaluates to the same bits"""
    c.note = ""
    rc = fn(c)
    err = c.lib.ea_last_error().decode()[:110]
    rc2, cost, J = c.ref()
    intact = rc2 == 0 and cost == c.ref_cost and np.array_equal(J, c.ref_J)
    msg = "rc=%s intact=%s %s | last_error: %s" % (rc, intact, c.note, err if rc not in (0, None) else "-")
    return ("ok" if (rc not in (0, None) and intact) else ("ACCEPTED" if intact else "HANDLE DAMAGED")), msg


def run_in_process():
    """every case on ONE context (what tests/test_gpu_misuse.py runs); -> list of (verdict, name, message)"""
    c = make_ctx()
    out = []
    for name, fn in cases():
        v, m = run_case(c, fn)
        out.append((v, name, m))
    return out


def run_child(name, fn, wfd):
    try:
        verdict, msg = run_case(make_ctx(), fn)
    except Exception:
        verdict, msg = "PYTHON EXCEPTION", traceback.format_exc().strip().splitlines()[-1]
    os.write(wfd, ("%s\t%s" % (verdict, msg)).encode())
    os._exit(0)


def main():
    results = []
    for name, fn in cases():
        if sel and sel not in name:
            continue
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            signal.alarm(90)
            run_child(name, fn, w)
        os.close(w)
        data = b""
        while True:
            chunk = os.read(r, 65536)
            if not chunk:
                break
            data += chunk
        os.close(r)
        _, status = os.waitpid(pid, 0)
        if os.WIFSIGNALED(status):
            verdict, msg = "CRASH", "signal %d%s" % (os.WTERMSIG(status), " (timeout)" if os.WTERMSIG(status) == signal.SIGALRM else "")
        else:
            verdict, _, msg = data.decode(errors="replace").partition("\t")
        results.append((verdict, name, msg))
        print("%-16s %-70s %s" % (verdict or "NO REPORT", name, msg), flush=True)
    bad = [r for r in results if r[0] not in ("ok", "ACCEPTED")]
    print("misuse probe: %d cases, %d refused cleanly, %d accepted (see notes), %d PROBLEMS" % (
        len(results), sum(r[0] == "ok" for r in results), sum(r[0] == "ACCEPTED" for r in results), len(bad)))


if __name__ == "__main__":
    main()

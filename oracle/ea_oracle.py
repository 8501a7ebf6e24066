"""ctypes binding of oracle/libea_oracle.so  (see oracle/ea_oracle.h).

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (no Ceres in this image; see the header of
ea_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libea_oracle.so")

LOSS_TRIVIAL, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2
JAC_ANALYTIC, JAC_JET = 0, 1
LIN_CHOLESKY, LIN_DENSE_QR = 0, 1
STRATEGY_LM, STRATEGY_DOGLEG = 0, 1
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
WHY = ["none", "function_tolerance", "gradient_tolerance", "parameter_tolerance",
       "max_iterations", "min_radius", "initial_eval_failed", "too_many_invalid_steps",
       "eval_failed"]
MAX_ITERS = 512


class Problem(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("grid", C.POINTER(C.c_double)), ("grid_rows", C.c_int), ("grid_cols", C.c_int),
                ("loss_kind", C.c_int), ("loss_a", C.c_double),
                ("z_guard", C.c_double), ("z_eps", C.c_double), ("rot_transposed", C.c_int),
                ("use_distortion", C.c_int),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double), ("k3", C.c_double),
                ("use_second_cam", C.c_int), ("T12", C.c_double * 16), ("T12inv", C.c_double * 16)]


class Term(C.Structure):
    _fields_ = [("problem", C.POINTER(Problem)), ("xyz", C.POINTER(C.c_double)), ("n", C.c_int64),
                ("stride", C.c_int)]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int),
                ("jacobi_scaling", C.c_int), ("jacobian_mode", C.c_int),
                ("linear_solver", C.c_int), ("strategy", C.c_int), ("verbose", C.c_int)]


class Summary(C.Structure):
    _fields_ = [("termination", C.c_int), ("why", C.c_int), ("num_iterations", C.c_int),
                ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_residual_evals", C.c_int64), ("num_jacobian_evals", C.c_int64),
                ("it_cost", C.c_double * MAX_ITERS), ("it_cost_change", C.c_double * MAX_ITERS),
                ("it_gradient_max_norm", C.c_double * MAX_ITERS),
                ("it_step_norm", C.c_double * MAX_ITERS),
                ("it_relative_decrease", C.c_double * MAX_ITERS),
                ("it_radius", C.c_double * MAX_ITERS), ("it_successful", C.c_int * MAX_ITERS)]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = [os.path.join(_HERE, f) for f in ("ea_oracle.c", "ea_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libea_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.ea_oracle_default_options.argtypes = [C.POINTER(Options)]
        L.ea_oracle_default_problem.argtypes = [C.POINTER(Problem)]
        L.ea_oracle_bicubic.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp]
        L.ea_oracle_block_jet.argtypes = [C.POINTER(Problem), dp, dp, dp, dp, dp, dp]
        L.ea_oracle_block_jet.restype = C.c_int
        L.ea_oracle_block_analytic.argtypes = [C.POINTER(Problem), dp, dp, dp, dp, dp]
        L.ea_oracle_block_analytic.restype = C.c_int
        L.ea_oracle_quat_plus.argtypes = [dp, dp, dp]
        L.ea_oracle_quat_plus_jacobian.argtypes = [dp, dp]
        L.ea_oracle_eval.argtypes = [C.POINTER(Problem), dp, C.c_int64, C.c_int, dp, dp, C.c_int,
                                     dp, dp, dp, dp, dp, dp, dp]
        L.ea_oracle_eval.restype = C.c_int64
        L.ea_oracle_cost.argtypes = [C.POINTER(Problem), dp, C.c_int64, C.c_int, dp, dp, dp]
        L.ea_oracle_cost.restype = C.c_int64
        L.ea_oracle_solve.argtypes = [C.POINTER(Problem), dp, C.c_int64, C.c_int,
                                      C.POINTER(Options), dp, dp, C.POINTER(Summary)]
        L.ea_oracle_solve.restype = C.c_int
        L.ea_oracle_eval_terms.argtypes = [C.POINTER(Term), C.c_int, dp, dp, C.c_int, dp, dp, dp]
        L.ea_oracle_eval_terms.restype = C.c_int64
        L.ea_oracle_solve_terms.argtypes = [C.POINTER(Term), C.c_int, C.POINTER(Options), dp, dp, C.POINTER(Summary)]
        L.ea_oracle_solve_terms.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleProblem:
    """Holds the grid + camera + loss the way the reference's set-up block does
    (standalone_edge_align.cpp:256-278)."""

    def __init__(self, grid, fx, fy, cx, cy, loss=LOSS_CAUCHY, loss_a=1.0,
                 z_guard=0.01, z_eps=0.0, rot_transposed=False, distortion=None, T12=None, T12inv=None):
        self.grid = _f64(grid)
        assert self.grid.ndim == 2
        p = Problem()
        lib().ea_oracle_default_problem(C.byref(p))
        p.fx, p.fy, p.cx, p.cy = fx, fy, cx, cy
        p.grid = _dp(self.grid)
        p.grid_rows, p.grid_cols = self.grid.shape
        p.loss_kind, p.loss_a = loss, loss_a
        p.z_guard, p.z_eps, p.rot_transposed = z_guard, z_eps, int(rot_transposed)
        if distortion is not None:   # (k1, k2, p1, p2, k3) in EAResidueEx::Create's order (utils.h:156-160)
            p.use_distortion = 1
            p.k1, p.k2, p.p1, p.p2, p.k3 = [float(x) for x in distortion]
        if T12 is not None:          # row-major 4x4, EAResidueSecondCam's ptrans_1to2 / ptrans_1to2_inv
            p.use_second_cam = 1
            T12 = _f64(T12).reshape(16)
            T12inv = _f64(T12inv if T12inv is not None else np.linalg.inv(T12.reshape(4, 4))).reshape(16)
            for i in range(16):
                p.T12[i] = T12[i]
                p.T12inv[i] = T12inv[i]
        self.p = p

    def bicubic(self, r, c):
        f, dr, dc = C.c_double(), C.c_double(), C.c_double()
        lib().ea_oracle_bicubic(self.p.grid, self.p.grid_rows, self.p.grid_cols, r, c,
                                C.byref(f), C.byref(dr), C.byref(dc))
        return f.value, dr.value, dc.value

    def block_jet(self, q, t, X):
        q, t, X = _f64(q), _f64(t), _f64(X)
        r = C.c_double()
        jq, jt = np.zeros(4), np.zeros(3)
        ok = lib().ea_oracle_block_jet(C.byref(self.p), _dp(q), _dp(t), _dp(X), C.byref(r),
                                       _dp(jq), _dp(jt))
        return bool(ok), r.value, jq, jt

    def block_analytic(self, q, t, X):
        q, t, X = _f64(q), _f64(t), _f64(X)
        r = C.c_double()
        j6 = np.zeros(6)
        ok = lib().ea_oracle_block_analytic(C.byref(self.p), _dp(q), _dp(t), _dp(X), C.byref(r),
                                            _dp(j6))
        return bool(ok), r.value, j6

    def eval(self, xyz, q, t, jacobian_mode=JAC_ANALYTIC, materialize=False):
        xyz, q, t = _f64(xyz), _f64(q), _f64(t)
        n, stride = xyz.shape
        cost = C.c_double()
        JtJ, Jtr = np.zeros((6, 6)), np.zeros(6)
        r = J = rr = rJ = None
        if materialize:
            r, J = np.zeros(n), np.zeros((n, 6))
            rr, rJ = np.zeros(n), np.zeros((n, 6))
        bad = lib().ea_oracle_eval(C.byref(self.p), _dp(xyz), n, stride, _dp(q), _dp(t),
                                   jacobian_mode, C.byref(cost), _dp(JtJ), _dp(Jtr),
                                   _dp(r), _dp(J), _dp(rr), _dp(rJ))
        out = dict(cost=cost.value, JtJ=JtJ, Jtr=Jtr, n_invalid=int(bad))
        if materialize:
            out.update(r=r, J=J, raw_r=rr, raw_J=rJ)
        return out

    def cost(self, xyz, q, t):
        xyz, q, t = _f64(xyz), _f64(q), _f64(t)
        n, stride = xyz.shape
        cost = C.c_double()
        bad = lib().ea_oracle_cost(C.byref(self.p), _dp(xyz), n, stride, _dp(q), _dp(t),
                                   C.byref(cost))
        return cost.value, int(bad)

    def solve(self, xyz, q, t, **opts):
        xyz = _f64(xyz)
        n, stride = xyz.shape
        q = _f64(q).copy()
        t = _f64(t).copy()
        o = Options()
        lib().ea_oracle_default_options(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise KeyError(k)
            setattr(o, k, v)
        s = Summary()
        lib().ea_oracle_solve(C.byref(self.p), _dp(xyz), n, stride, C.byref(o), _dp(q), _dp(t),
                              C.byref(s))
        ni = s.num_iterations + 1
        summary = dict(termination=s.termination, why=WHY[s.why], num_iterations=s.num_iterations,
                       num_successful_steps=s.num_successful_steps,
                       num_unsuccessful_steps=s.num_unsuccessful_steps,
                       initial_cost=s.initial_cost, final_cost=s.final_cost,
                       num_residual_evals=s.num_residual_evals,
                       num_jacobian_evals=s.num_jacobian_evals,
                       it_cost=np.array(s.it_cost[:ni]),
                       it_cost_change=np.array(s.it_cost_change[:ni]),
                       it_gradient_max_norm=np.array(s.it_gradient_max_norm[:ni]),
                       it_step_norm=np.array(s.it_step_norm[:ni]),
                       it_relative_decrease=np.array(s.it_relative_decrease[:ni]),
                       it_radius=np.array(s.it_radius[:ni]),
                       it_successful=np.array(s.it_successful[:ni]))
        return q, t, summary


def _make_terms(problems, clouds):
    arr = (Term * len(problems))()
    keep = []
    for i, (P, xyz) in enumerate(zip(problems, clouds)):
        xyz = _f64(xyz)
        keep.append(xyz)
        arr[i].problem = C.pointer(P.p)
        arr[i].xyz = _dp(xyz)
        arr[i].n, arr[i].stride = xyz.shape
    return arr, keep


def eval_terms(problems, clouds, q, t, jacobian_mode=JAC_ANALYTIC):
    """one ceres::Problem holding several residual families (e.g. camera 1 + camera 2) on one (q,t)"""
    arr, keep = _make_terms(problems, clouds)
    q, t = _f64(q), _f64(t)
    cost = C.c_double()
    JtJ, Jtr = np.zeros((6, 6)), np.zeros(6)
    bad = lib().ea_oracle_eval_terms(arr, len(problems), _dp(q), _dp(t), jacobian_mode, C.byref(cost), _dp(JtJ), _dp(Jtr))
    return dict(cost=cost.value, JtJ=JtJ, Jtr=Jtr, n_invalid=int(bad))


def solve_terms(problems, clouds, q, t, **opts):
    arr, keep = _make_terms(problems, clouds)
    q, t = _f64(q).copy(), _f64(t).copy()
    o = Options()
    lib().ea_oracle_default_options(C.byref(o))
    for k, v in opts.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    s = Summary()
    lib().ea_oracle_solve_terms(arr, len(problems), C.byref(o), _dp(q), _dp(t), C.byref(s))
    ni = s.num_iterations + 1
    return q, t, dict(termination=s.termination, why=WHY[s.why], num_iterations=s.num_iterations,
                      num_successful_steps=s.num_successful_steps, initial_cost=s.initial_cost,
                      final_cost=s.final_cost, it_cost=np.array(s.it_cost[:ni]), it_radius=np.array(s.it_radius[:ni]),
                      it_successful=np.array(s.it_successful[:ni]))


def quat_plus(q, delta):
    q, delta = _f64(q), _f64(delta)
    out = np.zeros(4)
    lib().ea_oracle_quat_plus(_dp(q), _dp(delta), _dp(out))
    return out


def quat_plus_jacobian(q):
    q = _f64(q)
    P = np.zeros((4, 3))
    lib().ea_oracle_quat_plus_jacobian(_dp(q), _dp(P))
    return P

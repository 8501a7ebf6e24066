"""Host-side cost of handing a frame pair over through the C-ABI the way the facade does per ceres::Solve: ea_problem_set_points
(AoS doubles -> SoA in the problem's dtype, upload) and ea_problem_set_dt (Grid2D view -> transposed, bordered image, upload), C2 size."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
cfg = synth.config_c2_twin(seed=2, n_points=50000)
for dtype, tag in ((capi.EA_F64, "f64"), (capi.EA_F32, "f32")):
    P = capi.Problem(*cfg["K"], dtype=dtype)
    xyz, grid = np.ascontiguousarray(cfg["xyz"]), np.ascontiguousarray(cfg["grid"])
    P.set_points(xyz); P.set_dt_grid(grid); P.eval([1, 0, 0, 0], [0, 0, 0])
    for name, fn in (("set_points 50000", lambda: P.set_points(xyz)), ("set_dt 640x480", lambda: P.set_dt_grid(grid)),
                     ("create + set_points + set_dt + solve + destroy", None)):
        ts = []
        for _ in range(30):
            t_ = time.perf_counter()
            if fn:
                fn()
            else:
                Q = capi.Problem(*cfg["K"], dtype=dtype); Q.set_points(xyz); Q.set_dt_grid(grid); Q.set_loss(capi.LOSS_CAUCHY, 1.0)
                Q.solve([1, 0, 0, 0], [0, 0, 0]); Q.close()
            ts.append(time.perf_counter() - t_)
        print("%s %-48s median %.3f ms  min %.3f ms" % (tag, name, np.median(ts) * 1e3, min(ts) * 1e3), flush=True)
    P.close()

# where a fresh problem's first solve spends its time
import collections
acc = collections.OrderedDict()
for rep in range(12):
    def lap(name, fn):
        t_ = time.perf_counter(); r = fn(); acc.setdefault(name, []).append(time.perf_counter() - t_); return r
    Q = lap("ea_problem_create", lambda: capi.Problem(*cfg["K"], dtype=capi.EA_F64))
    lap("set_points (first: 3 hipMalloc)", lambda: Q.set_points(xyz))
    lap("set_dt (first: hipMalloc)", lambda: Q.set_dt_grid(grid))
    lap("set_loss", lambda: Q.set_loss(capi.LOSS_CAUCHY, 1.0))
    lap("first solve (ea_batch_create inside)", lambda: Q.solve([1, 0, 0, 0], [0, 0, 0]))
    lap("second solve", lambda: Q.solve([1, 0, 0, 0], [0, 0, 0]))
    lap("destroy", lambda: Q.close())
for k, v in acc.items():
    print("  %-40s median %.3f ms  min %.3f ms" % (k, np.median(v[2:]) * 1e3, min(v) * 1e3))

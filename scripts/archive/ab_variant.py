"""kernel time of the variant (distortion + second camera) evaluation, for A/B runs (EA_HIP_LIB selects the build)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
K1, K2 = (525.0, 525.0, 319.5, 239.5), (520.0, 522.0, 321.0, 238.0)
DIST = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
T12 = synth.rigid_4x4(synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.04), [0.11, 0.004, -0.012])
Q = synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)); T = np.array([0.01, -0.005, 0.02])
fams = synth.make_stereo_problem(480, 640, 50000, 50000, 5, K1, K2, T12, Q, T, distortion=DIST)
tag = os.path.basename(os.environ.get('EA_HIP_LIB', 'current'))
for dtype in (capi.EA_F32, capi.EA_F64):
    for m in (1, 16):
        Ps = []
        for i in range(m):
            P1 = capi.Problem(*K1, dtype=dtype); P1.set_points(fams[0]["xyz"]); P1.set_dt_grid(fams[0]["grid"]); P1.set_distortion(*DIST)
            P2 = capi.Problem(*K2, dtype=dtype); P2.set_points(fams[1]["xyz"]); P2.set_dt_grid(fams[1]["grid"]); P2.set_distortion(*DIST); P2.set_second_camera(T12)
            P1.add_term(P2); Ps += [P1]
        B = capi.Batch(Ps)
        Qs = np.tile(Q, (m, 1)); Ts = np.tile(T, (m, 1))
        g = B.eval(Qs, Ts)
        k = min(B.bench_kernel(Qs, Ts, 5, 100) for _ in range(3))
        print("[%s] stereo Ex x%2d %s ppt %d rows %d | kernel(b2b) %.2f us | cost %.12g" % (tag, m, "f32" if dtype == capi.EA_F32 else "f64", B.info("points_per_thread"), B.info("num_tiles"), k * 1e3, g["cost"][0]), flush=True)
        B.close()

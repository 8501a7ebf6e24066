"""Generates tests/golden/ea_golden.npz — frozen oracle outputs on the reference's bundled frames.

Inputs are the reference's own data files (standalone/rgb-d/{rgb/1,3,5.png, depth/1.png}, copied
verbatim to tests/golden/rgbd/ — they are data, not source).  Expected outputs come from the
build-owned oracle (oracle/ea_oracle.c + oracle/preprocess_np.py): the reference itself cannot
be built here (Ceres/Eigen/OpenCV absent), so these vectors freeze the oracle, they do not pin
it to Ceres — PARITY UNPINNED, see oracle/ea_oracle.h.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ea_oracle as eo  # noqa: E402
from oracle import preprocess_np as pp  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
K = (525.0, 525.0, 319.5, 239.5)  # ref: standalone_edge_align.cpp:151-153

POSES = [
    (np.array([1.0, 0.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.0])),
    (np.array([0.99994526, 0.00880897, 0.00528135, -0.00199703]), np.array([0.00036988, -0.005963, -0.01456002])),
    (np.array([0.9990482, 0.0261769, -0.0348995, 0.0087265]), np.array([0.03, -0.02, 0.05])),
]


def load_pair(b):
    imA = pp.load_rgb_as_bgr(os.path.join(G, "rgbd", "rgb_1.png"))
    dA = pp.load_depth_u16(os.path.join(G, "rgbd", "depth_1.png"))
    imB = pp.load_rgb_as_bgr(os.path.join(G, "rgbd", "rgb_%d.png" % b))
    aX, _ = pp.get_aX(imA, dA, *K)
    dt = pp.get_distance_transform(imB)
    return aX, dt


def main():
    out = {}
    aX, dt3 = load_pair(3)
    _, dt5 = load_pair(5)
    out["n_points_frame1"] = np.int64(aX.shape[1])
    out["points_sum"] = aX[:3].sum(axis=1)
    out["dt3_sum"] = np.float64(dt3.astype(np.float64).sum())
    out["dt5_sum"] = np.float64(dt5.astype(np.float64).sum())
    for b, dt in ((3, dt3), (5, dt5)):
        grid = pp.grid_view_of_image(dt)
        P = eo.OracleProblem(grid, *K)
        for stride in (30, 1):
            X = aX[:3, ::stride].T.copy()
            tag = "b%d_s%d" % (b, stride)
            for k, (q, t) in enumerate(POSES):
                q = q / np.linalg.norm(q)
                e = P.eval(X, q, t, eo.JAC_ANALYTIC, materialize=True)
                out["%s_pose%d_q" % (tag, k)] = q
                out["%s_pose%d_t" % (tag, k)] = t
                out["%s_pose%d_cost" % (tag, k)] = np.float64(e["cost"])
                out["%s_pose%d_JtJ" % (tag, k)] = e["JtJ"]
                out["%s_pose%d_Jtr" % (tag, k)] = e["Jtr"]
                out["%s_pose%d_r64" % (tag, k)] = e["raw_r"][:64]
                out["%s_pose%d_J64" % (tag, k)] = e["raw_J"][:64]
            q, t, s = P.solve(X, [1, 0, 0, 0], [0, 0, 0])
            out["%s_lm_q" % tag] = q
            out["%s_lm_t" % tag] = t
            out["%s_lm_iterations" % tag] = np.int64(s["num_iterations"])
            out["%s_lm_successful" % tag] = np.int64(s["num_successful_steps"])
            out["%s_lm_why" % tag] = np.array(s["why"])
            out["%s_lm_it_cost" % tag] = s["it_cost"]
            out["%s_lm_it_radius" % tag] = s["it_radius"]
            print(tag, X.shape[0], s["why"], s["num_iterations"], s["initial_cost"], s["final_cost"], q, t)
    np.savez_compressed(os.path.join(G, "ea_golden.npz"), **out)
    print("wrote", os.path.join(G, "ea_golden.npz"), os.path.getsize(os.path.join(G, "ea_golden.npz")), "bytes")


if __name__ == "__main__":
    main()

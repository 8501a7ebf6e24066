"""Replay one case of scripts/soak.py (same seed, same draws) and print the solve traces of fp64 / fp32 GPU and the oracle."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
from oracle import ea_oracle as eo
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
cases = 0
while True:
    m = int(rng.integers(1, 7))
    dtype_f64 = rng.random() < 0.5
    loss = [(0, 1.0), (1, 1.0), (1, 0.3), (2, 0.2)][int(rng.integers(4))]
    specs = []
    for i in range(m):
        H, W = int(rng.integers(40, 260)), int(rng.integers(40, 340))
        n = int(rng.choice([0, 1, 63, 64, 65, 255, 256, 257, 1000, 5000, 20000, int(rng.integers(1, 30000))]))
        f = float(rng.uniform(0.6, 1.4) * W)
        args = (H, W, max(n, 8), int(rng.integers(4, 30)), int(rng.integers(1 << 30)), f, f * float(rng.uniform(0.95, 1.05)),
                (W - 1) / 2 + float(rng.normal()), (H - 1) / 2 + float(rng.normal()))
        pq = synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 1.5))); pt = tuple(rng.normal(size=3) * 0.01)
        norm = bool(rng.random() < 0.7)
        specs.append((args, pq, pt, norm, n))
    if rng.random() < 0.5:
        rng.choice([1, 2, 4]); rng.choice([256, 1024]); rng.integers(0, 2); rng.integers(0, 2)
    for i in range(m):
        rng.normal(size=3); rng.uniform(0, 2.0)
        if rng.random() < 0.25:
            rng.uniform(0.98, 1.02)
        rng.normal(size=3)
    idx = None
    if cases % 5 == 0:
        idx = int(rng.integers(m))
    if cases == want:
        break
    cases += 1
args, pq, pt, norm, n = specs[idx]
pr = synth.make_problem(*args, planted_q=pq, planted_t=pt, normalize=norm)
X = pr["xyz"][:n]
print("case", want, "problem", idx, "image", args[0], "x", args[1], "points", n, "K", pr["K"], "loss", loss, "normalize", norm)
O = eo.OracleProblem(pr["grid"], *pr["K"], loss=loss[0], loss_a=loss[1])
qo, to, so = O.solve(X, [1, 0, 0, 0], [0, 0, 0])
print("oracle   it", so["num_iterations"], so["why"], "cost", so["final_cost"], [ "%.9g" % c for c in so["it_cost"][:so["num_iterations"] + 1]])
for dt, name in ((capi.EA_F64, "gpu f64"), (capi.EA_F32, "gpu f32")):
    P = capi.Problem(*pr["K"], dtype=dt); P.set_points(X); P.set_dt_grid(pr["grid"]); P.set_loss(*loss)
    q, t, s = P.solve([1, 0, 0, 0], [0, 0, 0])
    print(name, " it", s["num_iterations"], s["why"], "cost", s["final_cost"], ["%.9g" % c for c in s["it_cost"][:s["num_iterations"] + 1]],
          "| dr %.2e dt %.2e vs oracle" % (synth.rotation_angle_between(q, qo), np.linalg.norm(t - to)))
    print("   successful", list(s["it_successful"][:s["num_iterations"] + 1]), "rel", ["%.3g" % c for c in s["it_relative_decrease"][:s["num_iterations"] + 1]])
    P.close()
print("planted pose error of the oracle's solve: %.2e rad" % synth.rotation_angle_between(qo, pq))

"""Randomised soak of the one-launch-per-iteration form of the solve (ea_lm_iter_kernel) against the (evaluate, step) pairs:
random image sizes, point counts from a handful up to the largest that qualifies (so that every launch shape the heuristics
pick for it occurs: 256 x 1 / 2 / 4), both dtypes, the three losses, LM and dogleg, iteration caps, picky acceptance
thresholds (rejected steps), far starts (large radii), non-unit start quaternions, the transposed-rotation flavour, batches of
1-4 problems inside the 256-workgroup limit, repeated solves on one batch.  Every solve must give the same bits in both
forms (poses, summaries, traces).  usage: python scripts/soak_fused.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if torch.cuda.is_available():
    torch.cuda.init()
from edge_alignment_amd import capi, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2025
rng = np.random.default_rng(seed)
t_end = time.time() + budget
TRACE = ("it_cost", "it_cost_change", "it_gradient_max_norm", "it_step_norm", "it_relative_decrease", "it_radius", "it_successful")
cases = fused_cases = rejected = capped = dogleg = shapes2 = shapes4 = 0
while time.time() < t_end:
    m = int(rng.choice([1, 1, 1, 2, 3, 4]))
    dtype = capi.EA_F64 if rng.random() < 0.5 else capi.EA_F32
    big = m == 1 and rng.random() < 0.25
    Ps, q0, t0 = [], [], []
    loss = [(0, 1.0), (1, 1.0), (1, 0.3), (2, 0.2)][int(rng.integers(4))]
    for i in range(m):
        if big:
            H, W = 480, 640
            n = int(rng.integers(60000, 131000 if dtype == capi.EA_F64 else 262000))
        else:
            H, W = int(rng.integers(60, 260)), int(rng.integers(80, 340))
            n = int(rng.choice([8, 63, 64, 65, 255, 256, 257, 1000, 5000, int(rng.integers(8, 16000))]))
        f = float(rng.uniform(0.7, 1.3) * W)
        pr = synth.make_problem(H, W, n, int(rng.integers(6, 60)), int(rng.integers(1 << 30)), f, f, (W - 1) / 2, (H - 1) / 2,
                                planted_q=synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.1, 2.5))),
                                planted_t=tuple(rng.normal(size=3) * 0.015), normalize=bool(rng.random() < 0.6))
        P = capi.Problem(*pr["K"], dtype=dtype)
        P.set_points(pr["xyz"]); P.set_dt_grid(pr["grid"]); P.set_loss(*loss)
        Ps.append(P)
    flavour = rng.random() < 0.15
    if flavour:
        for P in Ps:
            P.set_flavour(0.0, 0.001, 1)
    mode = int(rng.integers(4))
    for i in range(m):
        if mode == 0:
            q0.append([1.0, 0, 0, 0]); t0.append([0.0, 0, 0])
        elif mode == 1:
            q0.append(synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(5.0, 20.0)))); t0.append(rng.uniform(-0.3, 0.3, 3))
        elif mode == 2:
            q0.append(np.array([1.0, 0, 0, 0]) + 0.02 * rng.normal(size=4)); t0.append([0.0, 0, 0])   # non-unit
        else:
            q0.append([1.0, 0, 0, 0]); t0.append(rng.normal(size=3) * 0.01)
    opts = dict(max_num_iterations=int(rng.choice([2, 6, 25, 50])))
    if rng.random() < 0.3:
        opts["strategy"] = capi.STRATEGY_DOGLEG; dogleg += 1
    if rng.random() < 0.4:
        opts["min_relative_decrease"] = float(rng.choice([0.5, 0.9, 0.97]))
    if mode == 1:
        opts["initial_trust_region_radius"] = float(rng.choice([1e4, 1e8, 1e16]))
    B = capi.Batch(Ps)
    res = {}
    for rep in range(int(rng.choice([1, 1, 2]))):
        for fused in (1, 0):
            B.set_tuning("fused_iterations", -1 if fused else 0)
            q, t, s = B.solve(q0, t0, **opts)
            res[fused] = (q, t, s, B.info("fused_iterations"), B.info("points_per_thread"))
        a, b = res[1], res[0]
        where = (seed, cases, m, dtype, loss, opts, flavour, mode, [P.num_points for P in Ps])
        assert b[3] == 0, where
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), where
        for x, y in zip(a[2], b[2]):
            for k in ("termination", "why", "num_iterations", "num_successful_steps", "num_unsuccessful_steps", "initial_cost", "final_cost", "num_point_evals"):
                assert x[k] == y[k], (where, k)
            for k in TRACE:
                assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])), (where, k)
    cases += 1
    fused_cases += int(a[3] == 1)
    rejected += int(any(x["num_unsuccessful_steps"] > 0 for x in a[2]))
    capped += int(any(x["why"] == "max_iterations" for x in a[2]))
    shapes2 += int(a[3] == 1 and a[4] == 2); shapes4 += int(a[3] == 1 and a[4] == 4)
    B.close()
    for P in Ps:
        P.close()
print("soak (fused iterations) ok: %d cases, %d of them in the one-launch form (%d at two, %d at four points per lane), %d with rejected steps, %d cut off by the cap, %d dogleg, seed %d"
      % (cases, fused_cases, shapes2, shapes4, rejected, capped, dogleg, seed))

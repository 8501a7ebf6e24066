"""Randomised soak of the HIP path against the CPU oracle (test infrastructure): random image sizes, point counts (empty
included), losses, dtypes, poses (unit and non-unit quaternions), batches of 1-6 problems, every launch shape the
heuristics or the tuning keys can select; every few cases a full solve, the materialised rows against the fused sums, a pipelined sequence (riding fold).  usage: python scripts/soak.py [seconds] [seed]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
from oracle import ea_oracle as eo
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = solves = rows_checked = pipelines = pose_sets = mirrored = 0
worst = {"f64": 0.0, "f32": 0.0, "pose_f64": 0.0, "pose_f32": 0.0}
while time.time() < t_end:
    m = int(rng.integers(1, 7))
    dtype = capi.EA_F64 if rng.random() < 0.5 else capi.EA_F32
    tag = "f64" if dtype == capi.EA_F64 else "f32"
    loss = [(0, 1.0), (1, 1.0), (1, 0.3), (2, 0.2)][int(rng.integers(4))]
    Ps, Os, Xs, prs = [], [], [], []
    for i in range(m):
        H, W = int(rng.integers(40, 260)), int(rng.integers(40, 340))
        n = int(rng.choice([0, 1, 63, 64, 65, 255, 256, 257, 1000, 5000, 20000, int(rng.integers(1, 30000))]))
        f = float(rng.uniform(0.6, 1.4) * W)
        pr = synth.make_problem(H, W, max(n, 8), int(rng.integers(4, 30)), int(rng.integers(1 << 30)), f, f * float(rng.uniform(0.95, 1.05)),
                                (W - 1) / 2 + float(rng.normal()), (H - 1) / 2 + float(rng.normal()),
                                planted_q=synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 1.5))),
                                planted_t=tuple(rng.normal(size=3) * 0.01), normalize=bool(rng.random() < 0.7))
        X = pr["xyz"][:n]
        P = capi.Problem(*pr["K"], dtype=dtype); P.set_points(X); P.set_dt_grid(pr["grid"]); P.set_loss(*loss)
        Ps.append(P); Os.append(eo.OracleProblem(pr["grid"], *pr["K"], loss=loss[0], loss_a=loss[1])); Xs.append(X); prs.append(pr)
    B = capi.Batch(Ps)
    if rng.random() < 0.5:
        B.set_tuning("points_per_thread", int(rng.choice([1, 2, 4]))); B.set_tuning("threads", int(rng.choice([256, 1024])))
        B.set_tuning("buffer_loads", int(rng.integers(0, 2))); B.set_tuning("use_lds", int(rng.integers(0, 2)))
    if rng.random() < 0.25:
        B.set_tuning("wide_accumulate", 1)   # (in effect for plain fp32 batches on the L2 path, ignored otherwise)
    Q = np.zeros((m, 4)); T = np.zeros((m, 3))
    for i in range(m):
        q = synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 2.0)))
        if rng.random() < 0.25:
            q = q * rng.uniform(0.98, 1.02)
        Q[i] = q; T[i] = rng.normal(size=3) * 0.02
    g = B.eval(Q, T)
    tol = 1e-10 if dtype == capi.EA_F64 else 2e-4
    for i in range(m):
        if Xs[i].shape[0] == 0:
            assert g["cost"][i] == 0.0
            continue
        e = Os[i].eval(Xs[i], Q[i], T[i])
        assert int(g["n_invalid"][i]) == int(e["n_invalid"]), (cases, i)
        if e["n_invalid"]:
            continue
        if dtype == capi.EA_F32 and Xs[i].shape[0] < 64:
            continue  # a handful of points in fp32: a residual near zero has no relative accuracy to speak of (fp64 covers the indexing)
        slack = 100.0 if Xs[i].shape[0] <= 2 else 1.0
        def relerr(a, b):  # (an exactly flat sample -- a point outside the image -- has zero rows: absolute floor)
            a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
            return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)
        rel = max(relerr(g["cost"][i], e["cost"]), relerr(g["JtJ"][i], e["JtJ"]),
                  relerr(g["Jtr"][i], e["Jtr"]) if np.abs(e["Jtr"]).max() > 1e-6 * np.sqrt(np.abs(e["JtJ"]).max() * max(e["cost"], 1e-30)) else 0.0)
        # (fp32: the rounding of a point's projected coordinates moves its sample by ~1e-4 relative; a sum over fewer than
        # a thousand points does not average that away)
        lim = slack * tol * (5.0 if (dtype == capi.EA_F32 and Xs[i].shape[0] < 1000) else 1.0)
        assert rel <= lim, (cases, i, tag, rel, Xs[i].shape, loss)
        if slack == 1.0:
            worst[tag] = max(worst[tag], rel)
    # materialised mode: the rows of every point, in a random layout / store form, must sum to the fused mode's system
    # (any size, any launch shape), and their residuals must be the oracle's
    if cases % 2 == 0:
        layout = int(rng.integers(0, 2))
        B.set_tuning("rows_staged", int(rng.integers(0, 2))); B.set_tuning("rows_nontemporal", int(rng.integers(0, 2)))
        r, J, bad = B.eval_rows(Q, T, corrected=True, layout=layout)
        Jr = (J if layout == 0 else J.T).astype(np.float64)
        off = B.row_offsets()
        assert bad == int(g["n_invalid"].sum()), (cases, "rows: invalid count")
        for i in range(m):
            n = Xs[i].shape[0]
            assert off[i + 1] - off[i] == n, (cases, i)
            if n == 0 or g["n_invalid"][i] or (dtype == capi.EA_F32 and n < 64):
                continue
            ri, Ji = r[off[i]:off[i + 1]].astype(np.float64), Jr[off[i]:off[i + 1]]
            scale = max(np.abs(g["JtJ"][i]).max(), 1e-9)
            lim = (1e-10 if dtype == capi.EA_F64 else 2e-4) * (100.0 if n <= 2 else (5.0 if (dtype == capi.EA_F32 and n < 1000) else 1.0))
            assert np.abs(Ji.T @ Ji - g["JtJ"][i]).max() <= lim * scale, (cases, i, tag, "rows: JtJ", n)
            rows_checked += 1
    # the riding fold: a pipelined sequence of 2..5 steps ends on ea_batch_eval's sums (1e-13: another summation order),
    # riding and closing folds agree bit for bit
    if cases % 3 == 0 and B.info("lds_bytes") == 0 and B.info("wide_accumulate") == 0:
        k = int(rng.integers(2, 6))
        if cases % 2:
            B.bench_capture_pipelined(k)
            B.bench_steps(k)
        else:
            B.bench_steps(k, riding=True)   # the same launches without a graph
        last, riding = B.bench_result(), B.bench_result(riding=True)
        for key in ("cost", "JtJ", "Jtr"):
            assert np.array_equal(last[key], riding[key]), (cases, "riding != closing", key)
            assert np.abs(last[key] - g[key]).max() <= 1e-12 * max(np.abs(g[key]).max(), 1e-300), (cases, "riding fold vs eval", key)
        assert np.array_equal(last["n_invalid"], g["n_invalid"]), (cases, "riding fold: invalid count")
        pipelines += 1
    # the product form of the same thing (round 3): ea_batch_eval_poses with K different poses per problem -- every pose's
    # sums against ea_batch_eval at that pose (1e-12 fp64 / 2e-6 fp32: another summation order, pose constants built on the
    # device), the dt_f32 mirror on or off at random (bit-identical either way, checked where both are taken)
    if cases % 4 == 1:
        K = int(rng.integers(1, 6))
        Qk = np.stack([Q] * K); Tk = np.stack([T + 0.002 * rng.normal(size=T.shape) for _ in range(K)])
        B.set_tuning("dt_f32", -1 if rng.random() < 0.7 else 0)
        pk = B.eval_poses(Qk, Tk)
        ptol = 1e-12 if dtype == capi.EA_F64 else 2e-6
        for k in range(K):
            gk = B.eval(Qk[k], Tk[k])
            for key in ("cost", "JtJ", "Jtr"):
                assert np.abs(pk[key][k] - gk[key]).max() <= ptol * max(np.abs(gk[key]).max(), 1e-300), (
                    cases, "eval_poses vs eval", k, key, tag, pk[key][k], gk[key], Qk[k], [x.shape for x in Xs], B.info("points_per_thread"), B.info("threads"),
                    B.info("lds_bytes"), B.info("wide_accumulate"), B.info("poses_ride"), gk["n_invalid"])
            assert np.array_equal(pk["n_invalid"][k], gk["n_invalid"]), (cases, "eval_poses: invalid count", k)
        if dtype == capi.EA_F64 and B.info("dt_f32") == 1:
            B.set_tuning("dt_f32", 0)
            pk0 = B.eval_poses(Qk, Tk)
            for key in ("cost", "JtJ", "Jtr", "n_invalid"):
                assert np.array_equal(pk[key], pk0[key]), (cases, "fp32-stored image changed the bits", key)
            mirrored += 1
        B.set_tuning("dt_f32", -1)
        pose_sets += 1
    if cases % 5 == 0:
        i = int(rng.integers(m))
        if Xs[i].shape[0] >= 500:
            q1, t1, s1 = Ps[i].solve([1, 0, 0, 0], [0, 0, 0])
            qo, to, so = Os[i].solve(Xs[i], [1, 0, 0, 0], [0, 0, 0])
            if so["termination"] != 2 and s1["termination"] != 2:
                dr, dt = synth.rotation_angle_between(q1, qo), float(np.linalg.norm(t1 - to))
                lim = (1e-7, 1e-7) if dtype == capi.EA_F64 else (1e-4, 1e-3)
                # (a perfect fit -- trivial loss, cost down at 1e-24 -- ends on whichever tolerance rounding noise trips first:
                # seed 4242, case 49620 stops on the gradient tolerance in the oracle and one iteration later on the parameter
                # tolerance on the device, in both forms of the loop; the iteration count is compared while the cost means something)
                noise_floor = so["final_cost"] <= 1e-18 * max(so["initial_cost"], 1e-300)
                if dtype == capi.EA_F64 and not noise_floor and s1["num_iterations"] != so["num_iterations"]:
                    # keep what is needed to look at the case off-line (tests/ and scripts/ only: the oracle is the checker)
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez("gpurun_out/soak_solve_mismatch_%d_%d.npz" % (seed, cases), xyz=Xs[i], grid=prs[i]["grid"], K=np.array(prs[i]["K"]),
                             loss=np.array(loss), it_gpu=s1["it_cost"], it_oracle=so["it_cost"], ok_gpu=s1["it_successful"], ok_oracle=so["it_successful"])
                    Bs = capi.Batch([Ps[i]]); Bs.set_tuning("fused_iterations", 0)
                    qp, tp, sp = Bs.solve([1, 0, 0, 0], [0, 0, 0]); Bs.close()
                    print("solve mismatch, case %d problem %d (%d points, loss %s): gpu %d iterations (%s), pairs form %d (%s), oracle %d (%s)"
                          % (cases, i, Xs[i].shape[0], loss, s1["num_iterations"], s1["why"], sp[0]["num_iterations"], sp[0]["why"], so["num_iterations"], so["why"]))
                    print("  gpu    costs", ["%.12g" % c for c in s1["it_cost"]], list(s1["it_successful"]))
                    print("  oracle costs", ["%.12g" % c for c in so["it_cost"]], list(so["it_successful"]))
                    print("  gpu rel", ["%.6g" % c for c in s1["it_relative_decrease"]], "step", ["%.3g" % c for c in s1["it_step_norm"]])
                if dtype == capi.EA_F64 and not noise_floor:
                    assert s1["num_iterations"] == so["num_iterations"], (cases, i)
                # (fp32 on a few hundred points: the function-tolerance stop leaves more slack than the bar; report, assert
                # the bar from 5000 points up, where the bundled frames and every BASELINE config live)
                if dtype == capi.EA_F64 or Xs[i].shape[0] >= 5000:
                    assert dr < lim[0] and dt < lim[1], (cases, i, tag, dr, dt, Xs[i].shape, s1["num_iterations"], so["num_iterations"])
                elif dr >= lim[0]:
                    print("note: fp32 solve of %d points ends %.2e rad / %.2e m from the fp64 oracle (%d vs %d iterations)" % (
                        Xs[i].shape[0], dr, dt, s1["num_iterations"], so["num_iterations"]), flush=True)
                worst["pose_" + tag] = max(worst["pose_" + tag], dr)
                solves += 1
    B.close()
    for P in Ps:
        P.close()
    cases += 1
print("soak ok: %d batches, %d solves, %d row sets, %d pipelined sequences, %d K-pose calls (%d of them bit-compared with and without the float32 image), seed %d; worst relative sum error f64 %.2e f32 %.2e; worst pose difference f64 %.2e rad f32 %.2e rad" % (
    cases, solves, rows_checked, pipelines, pose_sets, mirrored, seed, worst["f64"], worst["f32"], worst["pose_f64"], worst["pose_f32"]))

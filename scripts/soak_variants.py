"""Randomised soak, second part: the residual variants (distortion, second camera, both), stereo problems whose two
families share one pose (in batches next to plain problems: the multi-term pose lookup of the evaluation kernel), DOGLEG
solves, the ROS flavour's knobs and the integer-pixel cost report -- GPU through the C-ABI against the CPU oracle /
numpy restatement.  usage: python scripts/soak_variants.py [seconds] [seed]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi, synth
from oracle import ea_oracle as eo, ea_numpy as en
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4321
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_cases = n_solves = n_fail = n_capped = n_split = 0
worst = {"f64": 0.0, "f32": 0.0}


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


while time.time() < t_end:
    H, W = int(rng.integers(60, 200)), int(rng.integers(80, 260))
    K1 = (float(rng.uniform(0.7, 1.2) * W), float(rng.uniform(0.7, 1.2) * W), (W - 1) / 2 + float(rng.normal()), (H - 1) / 2 + float(rng.normal()))
    K2 = tuple(k * float(rng.uniform(0.97, 1.03)) for k in K1)
    dist = tuple(rng.normal(size=5) * np.array([0.2, 0.4, 0.004, 0.004, 0.5])) if rng.random() < 0.6 else None
    T12 = synth.rigid_4x4(synth.quat_from_axis_angle(rng.normal(size=3), float(rng.uniform(0, 0.06))), rng.normal(size=3) * np.array([0.1, 0.01, 0.01]))
    Qp = synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.1, 1.2)))
    Tp = rng.normal(size=3) * 0.01
    n1, n2 = int(rng.integers(300, 6000)), int(rng.integers(300, 6000))
    fams = synth.make_stereo_problem(H, W, n1, n2, int(rng.integers(1 << 30)), K1, K2, T12, Qp, Tp, distortion=dist)
    loss = [(0, 1.0), (1, 1.0), (2, 0.3)][int(rng.integers(3))]
    dtype = capi.EA_F64 if rng.random() < 0.6 else capi.EA_F32
    tag, tol = ("f64", 1e-10) if dtype == capi.EA_F64 else ("f32", 5e-4)
    # A radial polynomial 1 + k1 r^2 + k2 r^4 + k3 r^6 with |k| > 0.5 cancels digits on points at normalised radius > 1.5
    # (the family's far points): fp32 rows of THOSE points are good to ~2e-3 (seed 4321, case 14048: k2 = -0.61).  The
    # relaxed bar applies to them alone (round 3): such a family is evaluated in two parts -- points inside radius 1.5 at
    # the ordinary 5e-4, the far points at 2e-3 -- so a regression on the well-conditioned rows stays visible.
    split_far = dtype == capi.EA_F32 and dist is not None and max(abs(dist[0]), abs(dist[1]), abs(dist[4])) > 0.5
    q = synth.quat_mul(synth.quat_from_axis_angle(rng.normal(size=3), float(rng.uniform(0, 0.01))), Qp)
    if rng.random() < 0.2:
        q = q * float(rng.uniform(0.98, 1.02))
    t = Tp + rng.normal(size=3) * 0.003
    kind = int(rng.integers(4))   # 0 plain camera 1, 1 second camera, 2 stereo pair (two terms), 3 stereo pair inside a batch
    O1 = eo.OracleProblem(fams[0]["grid"], *K1, loss=loss[0], loss_a=loss[1], distortion=dist)
    O2 = eo.OracleProblem(fams[1]["grid"], *K2, loss=loss[0], loss_a=loss[1], distortion=dist, T12=T12)

    def gpu(fam, K, second):
        P = capi.Problem(*K, dtype=dtype); P.set_points(fam["xyz"]); P.set_dt_grid(fam["grid"]); P.set_loss(*loss)
        if dist is not None: P.set_distortion(*dist)
        if second: P.set_second_camera(T12)
        return P
    P1, P2 = gpu(fams[0], K1, False), gpu(fams[1], K2, True)

    def far_mask(X, second):
        """normalised radius of every point in the camera that sees it, at pose (q, t) -- utils.h:120-127 / :208-245"""
        Rq = synth.quat_to_R(q)
        A = X
        if second:
            Ti = np.linalg.inv(T12)
            A = X @ Ti[:3, :3].T + Ti[:3, 3]
        Bp = A @ Rq.T + t
        if second:
            Bp = Bp @ T12[:3, :3].T + T12[:3, 3]
        with np.errstate(divide="ignore", invalid="ignore"):
            rn = np.hypot(Bp[:, 0] / Bp[:, 2], Bp[:, 1] / Bp[:, 2])
        return ~(rn <= 1.5)

    if split_far and kind < 2:
        # the two parts of the family, each against the oracle on the same points
        fam, Kc, second, Oc = (fams[1], K2, True, O2) if kind == 1 else (fams[0], K1, False, O1)
        far = far_mask(fam["xyz"], second)
        for part, ptol in ((~far, 5e-4), (far, 2e-3)):
            Xp = np.ascontiguousarray(fam["xyz"][part])
            if len(Xp) == 0:
                continue
            Pp = capi.Problem(*Kc, dtype=dtype); Pp.set_points(Xp); Pp.set_dt_grid(fam["grid"]); Pp.set_loss(*loss); Pp.set_distortion(*dist)
            if second: Pp.set_second_camera(T12)
            gp, ep = Pp.eval(q, t), Oc.eval(Xp, q, t, eo.JAC_JET)
            Pp.close()
            assert int(gp["n_invalid"]) == int(ep["n_invalid"]), (n_cases, kind, "part")
            if not ep["n_invalid"]:
                rp = max(rel(gp["cost"], ep["cost"]), rel(gp["JtJ"], ep["JtJ"]))
                if rp >= ptol:
                    print("FAIL case", n_cases, "kind", kind, tag, "part tol", ptol, "points", len(Xp), "rel", rp, "dist", dist)
                    n_fail += 1
                if ptol == 5e-4:
                    worst[tag] = max(worst[tag], rp)
        n_split += 1
        P1.close(); P2.close()
        n_cases += 1
        continue
    if split_far:
        tol = 2e-3   # (two-family kinds: the far points of both cameras sit in one sum; the per-part check above covers the family)
    if kind == 0:
        e = O1.eval(fams[0]["xyz"], q, t, eo.JAC_JET); g = P1.eval(q, t)
        got = (g["cost"], g["JtJ"], g["Jtr"], g["n_invalid"])
    elif kind == 1:
        e = O2.eval(fams[1]["xyz"], q, t, eo.JAC_JET); g = P2.eval(q, t)
        got = (g["cost"], g["JtJ"], g["Jtr"], g["n_invalid"])
    else:
        e = eo.eval_terms([O1, O2], [fams[0]["xyz"], fams[1]["xyz"]], q, t, eo.JAC_JET)
        P1.add_term(P2)
        if kind == 2:
            g = P1.eval(q, t); got = (g["cost"], g["JtJ"], g["Jtr"], g["n_invalid"])
        else:  # a plain problem in front of and behind the stereo pair: term index != group index for the later terms
            pr = synth.make_problem(H, W, 700, 9, int(rng.integers(1 << 30)), *K1, planted_q=Qp, planted_t=tuple(Tp), normalize=True)
            Pa = capi.Problem(*K1, dtype=dtype); Pa.set_points(pr["xyz"]); Pa.set_dt_grid(pr["grid"]); Pa.set_loss(*loss)
            Pb = capi.Problem(*K1, dtype=dtype); Pb.set_points(pr["xyz"][::2]); Pb.set_dt_grid(pr["grid"]); Pb.set_loss(*loss)
            Bt = capi.Batch([Pa, P1, Pb])
            qs = np.stack([q, q, q]); ts = np.stack([t, t, t])
            g = Bt.eval(qs, ts)
            ea = eo.OracleProblem(pr["grid"], *K1, loss=loss[0], loss_a=loss[1]).eval(pr["xyz"], q, t)
            eb = eo.OracleProblem(pr["grid"], *K1, loss=loss[0], loss_a=loss[1]).eval(pr["xyz"][::2], q, t)
            assert rel(g["cost"][0], ea["cost"]) < tol and rel(g["cost"][2], eb["cost"]) < tol, (n_cases, "neighbours of the stereo pair")
            got = (g["cost"][1], g["JtJ"][1], g["Jtr"][1], g["n_invalid"][1])
            Bt.close(); Pa.close(); Pb.close()
    assert int(got[3]) == int(e["n_invalid"]), (n_cases, kind)
    if not e["n_invalid"]:
        r = max(rel(got[0], e["cost"]), rel(got[1], e["JtJ"]))
        if r >= tol:
            print("FAIL case", n_cases, "kind", kind, tag, "rel", r, "dist", dist, "loss", loss, "n", n1, n2, "HxW", H, W, "K", K1)
            if kind < 2:
                Pk, Ok, fam = (P2, O2, fams[1]) if kind == 1 else (P1, O1, fams[0])
                rg, Jg = Pk.eval_points(q, t, corrected=False)
                em = Ok.eval(fam["xyz"], q, t, eo.JAC_JET, materialize=True)
                X = Pk.get_points()
                perm = None
                em2 = Ok.eval(X, q, t, eo.JAC_JET, materialize=True)
                dJ = np.abs(Jg - em2["raw_J"]).max(axis=1); dr = np.abs(rg - em2["raw_r"])
                idx = np.argsort(-dJ)[:5]
                for i in idx:
                    print("  pt", i, "X", X[i], "r gpu/or", rg[i], em2["raw_r"][i], "J gpu", Jg[i], "J or", em2["raw_J"][i])
                print("  max |dr|", dr.max(), "max |dJ|", dJ.max(), "max |J|", np.abs(em2["raw_J"]).max())
            n_fail += 1
        worst[tag] = max(worst[tag], r)
    if n_cases % 4 == 0 and kind >= 2 and dtype == capi.EA_F64:
        strat = capi.STRATEGY_DOGLEG if rng.random() < 0.5 else capi.STRATEGY_LM
        qo, to, so = eo.solve_terms([O1, O2], [fams[0]["xyz"], fams[1]["xyz"]], [1, 0, 0, 0], [0, 0, 0],
                                    strategy=eo.STRATEGY_DOGLEG if strat == capi.STRATEGY_DOGLEG else eo.STRATEGY_LM, max_num_iterations=40)
        q1, t1, s1 = P1.solve([1, 0, 0, 0], [0, 0, 0], strategy=strat, max_num_iterations=40)
        if so["termination"] != 2:
            n_solves += 1
            if so["termination"] == 0 and s1["termination"] == 0:
                ok = s1["num_iterations"] == so["num_iterations"] and synth.rotation_angle_between(q1, qo) < 1e-7 and np.linalg.norm(t1 - to) < 1e-7
            else:
                # a solve cut off by the iteration cap has not contracted onto a minimum: on the ill-conditioned members of
                # this family (strong distortion, radius 1e12 = plain Gauss-Newton on a near-singular system) rounding-level
                # differences in the sums grow by an order of magnitude per iteration (1e-15 at iteration 0, 1e-8 by
                # iteration 19 in the two cases that showed it).  What must hold: the same accept / reject pattern and the
                # same costs to 1e-9 over the first ten iterations, and an end no further from the oracle's than its own
                # last step moved it.
                m = min(10, len(s1["it_cost"]), len(so["it_cost"]))
                head = all(s1["it_successful"][k] == so["it_successful"][k] and abs(s1["it_cost"][k] - so["it_cost"][k]) <= 1e-9 * abs(so["it_cost"][k]) for k in range(m))
                ok = head and s1["termination"] == so["termination"] and abs(s1["final_cost"] - so["final_cost"]) <= 1e-3 * abs(so["final_cost"])
                # and, independent of where the two trajectories drifted to (round 3): the evaluator itself at the ORACLE's
                # final pose -- cost to 1e-9, the bar of every other fp64 evaluation
                ge = P1.eval(qo, to)
                eo_end = eo.eval_terms([O1, O2], [fams[0]["xyz"], fams[1]["xyz"]], qo, to, eo.JAC_JET)
                ok = ok and int(ge["n_invalid"]) == int(eo_end["n_invalid"]) and (eo_end["n_invalid"] or rel(ge["cost"], eo_end["cost"]) <= 1e-9)
                n_capped += 1
            if not ok:
                n_fail += 1
                print("FAIL solve case", n_cases, "strategy", strat, "dist", dist, "loss", loss, "iters", s1["num_iterations"], so["num_iterations"],
                      "dq", synth.rotation_angle_between(q1, qo), "dt", np.linalg.norm(t1 - to), "term", s1.get("termination"), so["termination"])
                print("  gpu costs", s1.get("initial_cost"), s1.get("final_cost"), "oracle", so.get("initial_cost"), so.get("final_cost"))
                for k in range(min(len(s1["it_cost"]), len(so["it_cost"]))):
                    print("   it %2d gpu cost %.15e radius %.6e ok %d | oracle %.15e %.6e %d" % (k, s1["it_cost"][k], s1["it_radius"][k], s1["it_successful"][k],
                                                                                     so["it_cost"][k], so["it_radius"][k], so["it_successful"][k]))
    if kind == 0 and dist is None and n_cases % 3 == 0:   # the integer-pixel report on the plain family
        X = P1.get_points()
        pc, want = P1.pixel_cost(q, t), en.pixel_cost(X, q, t, *K1, np.ascontiguousarray(fams[0]["grid"].T))
        assert pc["count"] == want["count"] and pc["outside"] == want["outside"] and pc["max_cost"] == want["max_cost"], n_cases
        assert rel(pc["total_cost"], want["total_cost"]) < 1e-12, n_cases
    P1.close(); P2.close()
    n_cases += 1
print("soak (variants) %s:" % ("ok" if not n_fail else "FAILED %d" % n_fail) + " %d cases, %d joint solves (%d cut off by the iteration cap), %d fp32 families with |k| > 0.5 checked in two parts, seed %d; worst relative error f64 %.2e f32 %.2e (fp32: rows inside normalised radius 1.5)" % (n_cases, n_solves, n_capped, n_split, seed, worst["f64"], worst["f32"]))

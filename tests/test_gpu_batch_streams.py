"""ea_batch_solve with concurrent sub-batches (tuning key "solve_streams"): every problem's solve is the one it gets in
the single-stream solve of the same batch -- same launch shape, same iterates, same summary -- whatever the number of
streams (the parts inherit the shape the whole batch resolves to)."""
import numpy as np
import pytest

from edge_alignment_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [capi.EA_F64, capi.EA_F32])
def test_concurrent_halves_solve_every_problem_identically(dtype):
    n = 18
    Ps = []
    for i in range(n):
        cfg = synth.config_c2_twin(seed=300 + i, n_points=3000 + 500 * (i % 5))
        P = capi.Problem(*cfg["K"], dtype=dtype)
        P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
        Ps.append(P)
    q0 = np.tile([1.0, 0, 0, 0], (n, 1)); t0 = np.zeros((n, 3))
    B = capi.Batch(Ps)
    try:
        out = {}
        for streams in (1, 2, 3, -1):
            B.set_tuning("solve_streams", streams)
            out[streams] = B.solve(q0, t0)
        q1, t1, s1 = out[1]
        assert all(s["num_iterations"] > 3 for s in s1)
        for streams in (2, 3, -1):
            q, t, s = out[streams]
            assert np.array_equal(q, q1) and np.array_equal(t, t1)
            for a, b in zip(s, s1):
                for k in ("why", "num_iterations", "num_successful_steps", "initial_cost", "final_cost"):
                    assert a[k] == b[k], (streams, k)
                assert np.array_equal(a["it_cost"], b["it_cost"])
        # and the single-problem solve of one of them
        qa, ta, sa = Ps[7].solve(q0[7], t0[7])
        assert np.array_equal(qa, q1[7]) and np.array_equal(ta, t1[7]) and sa["num_iterations"] == s1[7]["num_iterations"]
    finally:
        B.close()
        for P in Ps:
            P.close()


def test_random_batches_are_stream_invariant():
    """batch sizes / point counts around the thresholds of the launch-shape heuristics (they look at the batch's
    totals), both dtypes, a batch that mixes plain and distorted-camera problems"""
    rng = np.random.default_rng(12345)
    pool = {}

    def problem(dtype, i, distorted=False):
        key = (dtype, i, distorted)
        if key not in pool:
            cfg = synth.config_c2_twin(seed=500 + i, n_points=int(2000 + 700 * (i % 9)))
            P = capi.Problem(*cfg["K"], dtype=dtype)
            P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
            if distorted:
                P.set_distortion(1e-3, -2e-4, 1e-4, -1e-4, 0.0)
            pool[key] = P
        return pool[key]

    try:
        for trial in range(10):
            dtype = capi.EA_F64 if trial % 2 == 0 else capi.EA_F32
            n = int(rng.integers(16, 41))
            ids = rng.choice(50, n, replace=False)
            Ps = [problem(dtype, int(i), distorted=(trial >= 8 and k % 3 == 0)) for k, i in enumerate(ids)]
            B = capi.Batch(Ps)
            q0 = np.tile([1.0, 0, 0, 0], (n, 1)) + 0.003 * rng.standard_normal((n, 4))
            q0 /= np.linalg.norm(q0, axis=1)[:, None]
            t0 = 0.004 * rng.standard_normal((n, 3))
            ref = None
            for streams in (1, 2, 3, 1):
                B.set_tuning("solve_streams", streams)
                q, t, s = B.solve(q0, t0)
                sig = (q.tobytes(), t.tobytes(), tuple((x["why"], x["num_iterations"], x["final_cost"]) for x in s))
                if ref is None:
                    ref = sig
                assert sig == ref, (trial, streams)
            B.close()
    finally:
        for P in pool.values():
            P.close()


def test_handles_on_different_host_threads():
    """a handle is not thread-safe, different handles on different host threads are fine --
    four threads, each with its own problems and batch, solving at the same time, get what a lone thread gets."""
    import threading
    n_threads, per = 4, 6
    sets, refs = [], []
    for k in range(n_threads):
        Ps = []
        for i in range(per):
            cfg = synth.config_c2_twin(seed=900 + 10 * k + i, n_points=2500 + 300 * i)
            P = capi.Problem(*cfg["K"], dtype=capi.EA_F64)
            P.set_points(cfg["xyz"]); P.set_dt_grid(cfg["grid"]); P.set_loss(capi.LOSS_CAUCHY, 1.0)
            Ps.append(P)
        sets.append((Ps, capi.Batch(Ps)))
    q0 = np.tile([1.0, 0, 0, 0], (per, 1)); t0 = np.zeros((per, 3))
    try:
        for Ps, B in sets:
            q, t, s = B.solve(q0, t0)
            refs.append((q.tobytes(), t.tobytes(), tuple(x["num_iterations"] for x in s)))
        out = [None] * n_threads
        errs = []

        def work(k):
            try:
                res = []
                for _ in range(10):
                    q, t, s = sets[k][1].solve(q0, t0)
                    res.append((q.tobytes(), t.tobytes(), tuple(x["num_iterations"] for x in s)))
                    e = sets[k][0][0].eval(q0[0], t0[0])  # single-problem handle of the same thread in between
                    assert np.isfinite(e["cost"])
                out[k] = res
            except Exception as ex:  # surfaced below: an assertion in a thread would otherwise be lost
                errs.append(repr(ex))
        th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for k in range(n_threads):
            assert all(r == refs[k] for r in out[k]), k
    finally:
        for Ps, B in sets:
            B.close()
            for P in Ps:
                P.close()

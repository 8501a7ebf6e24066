"""Materialised mode (ea_eval_rows_kernel): kernel time, algorithmic GB/s (3 s in + 7 s out per point + one pass over the DT
image) and fraction of the 8 TB/s HBM roofline for C2, C5 and C2-shaped batches, both J layouts, LDS-staged / direct
row-major stores, plain / non-temporal stores.  Back-to-back launches between one event pair (ea_batch_bench_rows).

  python scripts/rows_sweep.py [quick]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # (PyTorch first: tests/conftest.py)
torch.cuda.init()
from edge_alignment_amd import capi, synth

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
HBM = 8000.0


def run(name, cfgs, dtype, tile=None, loss=(capi.LOSS_CAUCHY, 1.0)):
    es = 4 if dtype == capi.EA_F32 else 8
    Ps = []
    for c in cfgs:
        P = capi.Problem(*c["K"], dtype=dtype)
        if tile is not None:
            P.set_point_order(tile)
        P.set_points(c["xyz"]); P.set_dt_grid(c["grid"]); P.set_loss(*loss)
        Ps.append(P)
    B = capi.Batch(Ps)
    m = len(Ps)
    q, t = np.tile(np.array([1.0, 0, 0, 0]), (m, 1)), np.zeros((m, 3))
    n = sum(P.num_points for P in Ps)
    by = sum(10 * es * P.num_points + c["image"].shape[0] * c["image"].shape[1] * es for P, c in zip(Ps, cfgs))
    for layout in (0, 1):
        for mode in ((1, 0, 3, 2) if layout == 0 else (0, 2)):
            ms = min(B.bench_rows(q, t, 5, 50, corrected=True, layout=layout, mode=mode) for _ in range(3))
            print("%-24s %s points %8d | layout %d %-6s %-3s | %8.2f us  %7.0f GB/s  frac %.3f  %.3e evals/s" % (
                name, "f32" if es == 4 else "f64", n, layout, ("staged" if mode & 1 else "direct") if layout == 0 else "soa",
                "nt" if mode & 2 else "", ms * 1e3, by / (ms * 1e-3) / 1e9, by / (ms * 1e-3) / 1e9 / HBM, n / (ms * 1e-3)), flush=True)
    B.close()
    for P in Ps:
        P.close()


f64only = len(sys.argv) > 1 and sys.argv[1] == "f64only"
if f64only:
    batch = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
    run("c2", [synth.config_c2_twin(seed=2, n_points=50000)], capi.EA_F64)
    run("batch32 raster", batch, capi.EA_F64)
    run("batch32 tile16", batch, capi.EA_F64, tile=16)
    run("c5 fp64", [synth.config_c5()], capi.EA_F64, loss=(capi.LOSS_TRIVIAL, 1.0))
    sys.exit(0)
run("c5", [synth.config_c5()], capi.EA_F32, loss=(capi.LOSS_TRIVIAL, 1.0))
run("c2", [synth.config_c2_twin(seed=2, n_points=50000)], capi.EA_F64)
batch = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
run("batch32 raster", batch, capi.EA_F32)
run("batch32 tile16", batch, capi.EA_F32, tile=16)
if not quick:
    run("batch32 raster", batch, capi.EA_F64)
    run("batch32 tile16", batch, capi.EA_F64, tile=16)
    run("batch128 tile16", batch * 4, capi.EA_F32, tile=16)
    run("c5 fp64", [synth.config_c5()], capi.EA_F64, loss=(capi.LOSS_TRIVIAL, 1.0))

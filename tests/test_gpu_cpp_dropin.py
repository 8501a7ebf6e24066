"""The C++ drop-in surface on the GPU: the reference's own call sequence
(standalone/standalone_edge_align.cpp:256-293 and src/ea.cpp:184-191) compiled against
edge_alignment_amd/include/{EAResidue.h, ceres/*, ros/SolveEA.h} and linked to libea_hip.so."""
import os
import struct
import subprocess

import numpy as np
import pytest

from edge_alignment_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binaries(hip):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    return os.path.join(ROOT, "examples")


def _write_problem(path, aX, grid, K):
    # grid is the Grid2D view (W x H row-major) == the column-major H x W Eigen matrix's raw data
    W, H = grid.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", aX.shape[1], H, W))
        f.write(struct.pack("<dddd", *K))
        f.write(np.ascontiguousarray(aX.T, dtype=np.float64).tobytes())   # 4xN column-major
        f.write(np.ascontiguousarray(grid, dtype=np.float64).tobytes())


def test_edge_align_test1_call_sequence(binaries, bundled_pair, golden, tmp_path):
    p = str(tmp_path / "pair13.bin")
    _write_problem(p, bundled_pair["aX"], bundled_pair["grids"][3], bundled_pair["K"])
    for stride in (30, 1):
        out = subprocess.run([os.path.join(binaries, "standalone_test1"), p, str(stride)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        v = [float(x) for x in out.stdout.split()]
        q, t, iters, term = np.array(v[:4]), np.array(v[4:7]), int(v[7]), int(v[8])
        tag = "b3_s%d" % stride
        assert term == 0  # ceres::CONVERGENCE
        assert synth.rotation_angle_between(q, golden[tag + "_lm_q"]) < 1e-7
        assert np.linalg.norm(t - golden[tag + "_lm_t"]) < 1e-7
        # Ceres' "Minimizer iterations" = successful + unsuccessful steps; the last iteration, which
        # trips the function tolerance, is neither
        assert iters == int(golden[tag + "_lm_successful"])
        assert v[9] == pytest.approx(float(golden[tag + "_lm_it_cost"][0]), rel=1e-10)
        assert "Residual blocks" in out.stderr and "CONVERGENCE" in out.stderr


def test_solve_ea_class_dogleg(binaries, oracle, bundled_pair, tmp_path):
    p = str(tmp_path / "pair13.bin")
    _write_problem(p, bundled_pair["aX"], bundled_pair["grids"][3], bundled_pair["K"])
    out = subprocess.run([os.path.join(binaries, "solve_ea_demo"), p], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    v = [float(x) for x in out.stdout.strip().split("\n")[-1].split()]
    q, t = np.array(v[:4]), np.array(v[4:7])
    # SolveEA() hard-codes half-resolution TUM intrinsics, no loss, R^T, z+0.001, DOGLEG, 25 iterations
    O = oracle.OracleProblem(bundled_pair["grids"][3], 262.5, 262.5, 159.75, 119.75, loss=oracle.LOSS_TRIVIAL,
                             z_guard=0.0, z_eps=0.001, rot_transposed=True)
    X = bundled_pair["aX"][:3, ::10].T.copy()
    qo, to, so = O.solve(X, [1, 0, 0, 0], [0, 0, 0], strategy=oracle.STRATEGY_DOGLEG, max_num_iterations=25)
    assert synth.rotation_angle_between(q, qo) < 1e-7 and np.linalg.norm(t - to) < 1e-7
    assert int(v[7]) == so["termination"]

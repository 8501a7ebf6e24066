"""The drop-in functors keep the reference's `template <typename T> bool operator()(const T*, const T*, T*) const`
(standalone/utils.h:47-48, include/EAResidue.h:85-86): compiled with g++ against the shim headers, instantiated for
double and for ceres::Jet<double, 7>, and checked against the oracle's dual-number evaluation of the same functor text
(value, d r / d q, d r / d t) -- a CPU-only test of the API surface; the solver itself never evaluates on the host."""
import os
import subprocess

import numpy as np
import pytest

from edge_alignment_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(name, tmp_path):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "edge_alignment_amd", "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", name + ".cpp")])
    return exe


def _cases(rng, n, pr):
    rows = []
    for i in range(n):
        q = synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 3)))
        if i % 3 == 2:
            q = q * rng.uniform(0.97, 1.03)   # non-unit quaternion: the functor text does not normalise (standalone)
        t = rng.normal(size=3) * 0.02
        X = pr["xyz"][rng.integers(pr["xyz"].shape[0])]
        rows.append((q, t, X))
    rows.append((np.array([1.0, 0, 0, 0]), np.zeros(3), np.array([0.1, 0.2, 0.004])))   # inside the z guard
    return rows


def _run(exe, pr, cases):
    g = pr["grid"]
    lines = ["%d %d" % g.shape, " ".join("%.17g" % v for v in g.ravel()), "%.17g %.17g %.17g %.17g" % tuple(pr["K"]), str(len(cases))]
    for q, t, X in cases:
        lines.append(" ".join("%.17g" % v for v in list(q) + list(t) + list(X[:3])))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
    return [np.array(l.split(), dtype=float) for l in out.strip().splitlines()]


@pytest.fixture(scope="module")
def small_problem():
    return synth.make_problem(60, 80, 400, 12, 5, 65.0, 65.0, 39.5, 29.5,
                              planted_q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(0.7)), planted_t=(0.004, -0.002, 0.003),
                              normalize=True)


def test_standalone_functor_template_double_and_jet(tmp_path, oracle, small_problem):
    pr = small_problem
    exe = _build("functor_probe_standalone", tmp_path)
    cases = _cases(np.random.default_rng(11), 25, pr)
    got = _run(exe, pr, cases)
    O = oracle.OracleProblem(pr["grid"], *pr["K"])
    for (q, t, X), row in zip(cases, got):
        ok, r, jq, jt = O.block_jet(q, t, X[:3])
        assert bool(row[0]) == ok and bool(row[1]) == ok
        if not ok:
            continue
        # <double> and the value part of <Jet> (the two differ in the last bits: x / z against x * (1 / z))
        assert row[2] == pytest.approx(r, rel=1e-12, abs=1e-14) and row[3] == pytest.approx(r, rel=1e-12, abs=1e-14)
        want = np.concatenate([jq, jt])
        assert np.abs(row[4:11] - want).max() <= 1e-10 * max(1.0, np.abs(want).max())
    assert not bool(got[-1][0])   # utils.h:70-73 `return false`


def test_ros_functor_template_double_and_jet(tmp_path, oracle, small_problem):
    pr = small_problem
    exe = _build("functor_probe_ros", tmp_path)
    cases = _cases(np.random.default_rng(12), 25, pr)[:-1]
    got = _run(exe, pr, cases)
    # the oracle's knobs for this flavour: R transposed, divisor z + 0.001, no guard (include/EAResidue.h:90-105)
    O = oracle.OracleProblem(pr["grid"], *pr["K"], z_guard=0.0, z_eps=0.001, rot_transposed=True)
    for (q, t, X), row in zip(cases, got):
        qn = q / np.linalg.norm(q)
        assert bool(row[0]) and bool(row[1]) and row[3] == pytest.approx(row[2], rel=1e-12, abs=1e-14)
        # value: QuaternionToRotation normalises, so the functor sees the unit quaternion's rotation
        ok, r, jq, jt = O.block_jet(qn, t, X[:3])
        assert row[2] == pytest.approx(r, rel=1e-11, abs=1e-13)
        # derivatives by central differences of the <double> instantiation itself
        h = 1e-6
        num = np.zeros(7)
        for k in range(7):
            d = np.zeros(7); d[k] = h
            cp = [(q + d[:4], t + d[4:], X)]; cm = [(q - d[:4], t - d[4:], X)]
            num[k] = (_run(exe, pr, cp)[0][2] - _run(exe, pr, cm)[0][2]) / (2 * h)
        assert np.abs(row[4:11] - num).max() <= 2e-5 * max(1.0, np.abs(num).max())


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_variant_functors_through_cost_function_evaluate(tmp_path, oracle, variant):
    """EAResidueEx / EAResidueSecondCam / EAResidueSecondCamEx (utils.h:102-421) keep their templated call operators too,
    and the facade's AutoDiffCostFunction::Evaluate differentiates them with Jet<double, 7> as Ceres would: value and both
    Jacobian blocks against the oracle's dual-number restatement of the same functors (an independently written text)."""
    K1, K2 = (130.0, 132.0, 79.5, 59.5), (128.0, 129.0, 81.0, 58.0)
    dist = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633) if variant & 1 else None
    T12 = synth.rigid_4x4(synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.04), [0.11, 0.004, -0.012])
    Q, T = synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0)), np.array([0.01, -0.005, 0.02])
    fams = synth.make_stereo_problem(60, 80, 300, 300, 5, K1, K2, T12, Q, T, distortion=dist)
    fam, K = (fams[1], K2) if variant & 2 else (fams[0], K1)
    O = oracle.OracleProblem(fam["grid"], *K, distortion=dist, T12=T12 if variant & 2 else None)
    exe = _build("functor_probe_variants", tmp_path)
    rng = np.random.default_rng(20 + variant)
    cases = []
    for i in range(20):
        q = synth.quat_mul(synth.quat_from_axis_angle(rng.normal(size=3), 0.01), Q) * (rng.uniform(0.98, 1.02) if i % 3 == 2 else 1.0)
        cases.append((q, T + rng.normal(size=3) * 0.003, fam["xyz"][rng.integers(fam["xyz"].shape[0])]))
    g = fam["grid"]
    lines = ["%d %d" % g.shape, " ".join("%.17g" % v for v in g.ravel()), "%.17g %.17g %.17g %.17g %d" % (tuple(K) + (variant,)),
             " ".join("%.17g" % v for v in (dist or (0,) * 5)), " ".join("%.17g" % v for v in T12.ravel()),
             " ".join("%.17g" % v for v in np.linalg.inv(T12).ravel()), str(len(cases))]
    for q, t, X in cases:
        lines.append(" ".join("%.17g" % v for v in list(q) + list(t) + list(X[:3])))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
    rows = [np.array(l.split(), dtype=float) for l in out.strip().splitlines()]
    assert len(rows) == len(cases)
    for (q, t, X), row in zip(cases, rows):
        ok, r, jq, jt = O.block_jet(q, t, X[:3])
        assert ok and bool(row[0]) and bool(row[1])
        assert row[2] == pytest.approx(r, rel=1e-11, abs=1e-13) and row[3] == pytest.approx(r, rel=1e-11, abs=1e-13)
        want = np.concatenate([jq, jt])
        assert np.abs(row[4:11] - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


def test_facade_surface_links_and_runs_without_a_gpu(tmp_path):
    """Every name the reference's headers pull out of ceres:: (`using ceres::LossFunctionWrapper;` ...), the loss classes'
    Evaluate, wrapper ownership, the parameterisation's Plus / ComputeJacobian, the option and summary members its
    drivers touch: compiled -Wall -Werror against the facade, linked with libea_hip.so, run on the CPU."""
    from edge_alignment_amd import capi
    lib_dir = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "facade_surface")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "edge_alignment_amd", "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "facade_surface.cpp"),
                           "-L", lib_dir, "-lea_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "facade surface: ok" in out.stdout, out.stdout + out.stderr

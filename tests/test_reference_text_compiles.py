"""The reference's OWN solve-block text against the drop-in headers (VERDICT r02 "what's missing" 2).

INTEGRATION.md says the "Setup non-linear Least Squares" blocks of the reference compile unchanged against
edge_alignment_amd/include.  Here the literal lines are read from /root/reference AT TEST TIME (nothing of them is stored
in this repository), wrapped in the declarations they rely on from the rest of their translation unit -- a 30-line
stand-in for the three Eigen types they touch (Eigen is not in this image), the PoseManipUtils prototypes, the
`using` lines of the reference's own headers -- and handed to `g++ -fsyntax-only`.  What is checked is the boundary:
every name, overload and conversion those lines use exists in the facade with a compatible signature.

Skipped where /root/reference does not exist (the GPU box)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
INC = os.path.join(ROOT, "edge_alignment_amd", "include")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")

# The slice of Eigen the blocks use: MatrixXd (.data() / .rows() / .cols() / (i, j), column-major), Matrix4d, Matrix3d.
EIGEN_STANDIN = r"""
namespace Eigen {
template <int R, int C> struct StandInMatrix {
  double *d; int r, c;
  StandInMatrix() : d(nullptr), r(R < 0 ? 0 : R), c(C < 0 ? 0 : C) {}
  double *data() { return d; }
  const double *data() const { return d; }
  int rows() const { return r; }
  int cols() const { return c; }
  double &operator()(int i, int j) { return d[(long)j * r + i]; }
  const double &operator()(int i, int j) const { return d[(long)j * r + i]; }
};
typedef StandInMatrix<-1, -1> MatrixXd;
typedef StandInMatrix<4, 4> Matrix4d;
typedef StandInMatrix<3, 3> Matrix3d;
}
"""


def _lines(path, first, last):
    with open(path, errors="replace") as f:
        src = f.read().splitlines()
    return src[first - 1:last]


def _syntax_only(tmp_path, name, text, include_dirs):
    cpp = tmp_path / name
    cpp.write_text(text)
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall"]   # (stricter than the reference's own build, which passes -fpermissive)   # (-fpermissive: the reference's own flag, standalone/CMakeLists.txt:16)
    for d in include_dirs:
        cmd += ["-I", d]
    out = subprocess.run(cmd + [str(cpp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-4000:]


def test_standalone_test1_solve_block_literal(tmp_path):
    """standalone/standalone_edge_align.cpp:256-300: Grid2D view, interpolator, eigenmat_to_raw, the residual-block loop with
    EAResidue::Create and an unqualified `new CauchyLoss(1.)`, QuaternionParameterization, options, unqualified
    `Solver::Summary`, ceres::Solve, FullReport, raw_to_eigenmat."""
    path = os.path.join(REF, "standalone", "standalone_edge_align.cpp")
    block = _lines(path, 256, 300)
    assert "a_X" in block[0] and "raw_to_eigenmat" in block[-1], "the cited line range moved"
    body = "\n".join(block)
    for needle in ("ceres::Grid2D<double,1> grid(", "EAResidue::Create(", "new CauchyLoss(1.)", "Solver::Summary summary;",
                   "ceres::Solve( options, &problem, &summary );", "summary.FullReport()"):
        assert needle in body
    # the head of the reference's translation unit, as far as the block depends on it (:1-25): std / ceres namespaces
    # opened, <chrono>; utils.h (the functors) is what edge_alignment_amd/include/EAResidue.h replaces
    text = "\n".join([
        "#include <iostream>", "#include <chrono>", "#include <string>", "using namespace std;", EIGEN_STANDIN,
        "#include <ceres/ceres.h>", "#include <ceres/cubic_interpolation.h>", "#include <ceres/loss_function.h>",
        "using namespace ceres;", '#include "EAResidue.h"',
        "struct PoseManipUtils {",   # standalone/PoseManipUtils.h:16-17
        "  static void raw_to_eigenmat( const double * quat, const double * t, Eigen::Matrix4d& dstT );",
        "  static void eigenmat_to_raw( const Eigen::Matrix4d& T, double * quat, double * t);",
        "};",
        # the locals of edge_align_test1 the block reads (:152, :169, :205, :240)
        "int solve_block(Eigen::MatrixXd &a_X, Eigen::MatrixXd &e_disTrans, Eigen::Matrix4d &b_T_a_optvar) {",
        "  double fx = 525., fy = 525., cx = 319.5, cy = 239.5;",
        body,
        "  return 0;", "}", ""])
    _syntax_only(tmp_path, "standalone_block.cpp", text, [INC])


def test_ros_set_as_ceres_problem_literal(tmp_path):
    """src/SolveEA.cpp:124-216 (the whole SolveEA::setAsCERESProblem) as a member of a class with the reference's own data
    members (include/SolveEA.h:51-64), behind the `using` lines of the reference's headers (include/SolveEA.h:21-32,
    include/EAResidue.h:25-35), against ros/EAResidue.h."""
    path = os.path.join(REF, "src", "SolveEA.cpp")
    block = _lines(path, 124, 216)
    assert block[0].startswith("void SolveEA::setAsCERESProblem()") and block[-1].strip() == "}", "the cited line range moved"
    body = "\n".join(block)
    for needle in ("ceres::Grid2D<double,2> grid(", "BiCubicInterpolator< Grid2D<double,2> >", "new AutoDiffCostFunction<EAResidue,1,4,3>",
                   "new ceres::HuberLoss(0.1)", "options.trust_region_strategy_type = ceres::DOGLEG;", "Solve(options, &problem, &summary);"):
        assert needle in body
    usings = [l for l in _lines(os.path.join(REF, "include", "SolveEA.h"), 21, 32) + _lines(os.path.join(REF, "include", "EAResidue.h"), 25, 36)
              if re.match(r"\s*using\s", l) and "cv" not in l]
    assert any("using namespace ceres;" in l for l in usings) and any("using ceres::Solve;" in l for l in usings)
    text = "\n".join([
        "#include <iostream>", "#include <cmath>", EIGEN_STANDIN,
        "#include <ceres/ceres.h>", "#include <ceres/loss_function.h>", "#include <ceres/local_parameterization.h>",
        "#include <ceres/rotation.h>", "#include <ceres/cubic_interpolation.h>",
        "#include <EAResidue.h>"] + usings + [
        "class SolveEA {", "public:", "  void setAsCERESProblem();", "private:",
        "  Matrix3d K;", "  MatrixXd now_dist_transform_eig;", "  MatrixXd list_edge_ref;", "};",
        body, ""])
    _syntax_only(tmp_path, "ros_block.cpp", text, [os.path.join(INC, "ros"), INC])


def test_shipped_ros_header_keeps_the_reference_class_surface():
    """include/SolveEA.h:36-47: the six public members src/ea.cpp:184-191 calls exist in the drop-in with the same names"""
    ref = "\n".join(_lines(os.path.join(REF, "include", "SolveEA.h"), 36, 47))
    names = re.findall(r"\b(SolveEA|setRefFrame|setNowFrame|setAsCERESProblem|_verify3dPts|_sampleCERESProblem)\s*\(", ref)
    assert set(names) == {"SolveEA", "setRefFrame", "setNowFrame", "setAsCERESProblem", "_verify3dPts", "_sampleCERESProblem"}
    ours = open(os.path.join(INC, "ros", "SolveEA.h")).read()
    for n in set(names):
        assert re.search(r"\b%s\s*\(" % re.escape(n), ours), n

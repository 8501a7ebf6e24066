// Source-level surface of the ceres:: facade, as far as the reference's drivers name it (using-declarations of
// include/EAResidue.h:24-31 and include/SolveEA.h:22-29; option / summary members of standalone_edge_align.cpp:282-293 and
// src/SolveEA.cpp:184-213; loss classes of :272, :2604 and src/SolveEA.cpp:144).  Compiled with -Wall -Werror, linked against
// libea_hip.so, run WITHOUT a GPU: only what needs no device is executed (loss formulas, wrapper ownership, the
// parameterisation's Plus / Jacobian, an empty Solve's summary).
#include <cmath>
#include <cstdio>

#include "EAResidue.h"
#include "ceres/ceres.h"

using ceres::AutoDiffCostFunction;
using ceres::CostFunction;
using ceres::LocalParameterization;
using ceres::LossFunction;
using ceres::LossFunctionWrapper;
using ceres::Problem;
using ceres::Solve;
using ceres::Solver;

static int g_deleted = 0;
struct CountedLoss : ceres::HuberLoss {
  CountedLoss() : ceres::HuberLoss(0.1) {}
  ~CountedLoss() override { ++g_deleted; }
};

static int check(bool ok, const char *what) {
  if (!ok) std::printf("FAIL %s\n", what);
  return ok ? 0 : 1;
}

int main() {
  int bad = 0;
  double rho[3];
  // ceres loss_function.cc: Trivial, Cauchy(a): b = a^2, rho = b log(1 + s/b); Huber(a): s <= a^2 ? s : 2 a sqrt(s) - a^2
  ceres::TrivialLoss().Evaluate(2.5, rho);
  bad += check(rho[0] == 2.5 && rho[1] == 1.0 && rho[2] == 0.0, "trivial");
  ceres::CauchyLoss(0.5).Evaluate(0.75, rho);
  bad += check(std::fabs(rho[0] - 0.25 * std::log(4.0)) < 1e-15 && std::fabs(rho[1] - 0.25) < 1e-15 && std::fabs(rho[2] + 4.0 / 16.0) < 1e-15, "cauchy");
  ceres::HuberLoss(0.1).Evaluate(0.04, rho);
  bad += check(std::fabs(rho[0] - (2 * 0.1 * 0.2 - 0.01)) < 1e-15 && std::fabs(rho[1] - 0.5) < 1e-15 && std::fabs(rho[2] + 0.5 / 0.08) < 1e-13, "huber outlier");
  ceres::HuberLoss(0.1).Evaluate(0.005, rho);
  bad += check(rho[0] == 0.005 && rho[1] == 1.0 && rho[2] == 0.0, "huber inlier");
  {
    LossFunctionWrapper w(new CountedLoss, ceres::TAKE_OWNERSHIP);
    bad += check(w.ea_kind() == EA_LOSS_HUBER && w.ea_scale() == 0.1, "wrapper forwards kind and scale");
    w.Reset(new ceres::CauchyLoss(2.0), ceres::TAKE_OWNERSHIP);
    bad += check(g_deleted == 1 && w.ea_kind() == EA_LOSS_CAUCHY && w.ea_scale() == 2.0, "wrapper reset deletes what it owned");
    CountedLoss mine;
    w.Reset(&mine, ceres::DO_NOT_TAKE_OWNERSHIP);
    w.Evaluate(0.04, rho);
    bad += check(std::fabs(rho[1] - 0.5) < 1e-15, "wrapper evaluates the wrapped loss");
    w.Reset(NULL, ceres::TAKE_OWNERSHIP);
    w.Evaluate(3.0, rho);
    bad += check(g_deleted == 1 && rho[0] == 3.0 && w.ea_kind() == EA_LOSS_TRIVIAL, "wrapper around NULL is trivial");
  }
  bad += check(g_deleted == 2, "borrowed loss destroyed once, by its owner");

  // Problem ownership (Ceres' default): cost functions and losses handed to AddResidualBlock die with the problem, a loss
  // shared by several blocks -- consecutive or not -- exactly once
  {
    g_deleted = 0;
    double img[16 * 12];
    for (int i = 0; i < 16 * 12; ++i) img[i] = 0.01 * i;
    ceres::Grid2D<double, 1> grid(img, 0, 16, 0, 12);
    ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interp(grid);
    double qq[4] = {1, 0, 0, 0}, tt[3] = {0, 0, 0};
    {
      Problem problem;
      CountedLoss *shared_a = new CountedLoss, *shared_b = new CountedLoss;
      LossFunction *order[6] = {shared_a, shared_a, shared_b, shared_a, new CountedLoss, shared_b};
      for (int i = 0; i < 6; ++i)
        problem.AddResidualBlock(EAResidue::Create(10., 10., 7.5, 5.5, 0.1 * i, 0.05 * i, 1.0 + i, interp), order[i], qq, tt);
      problem.AddResidualBlock(EAResidue::Create(10., 10., 7.5, 5.5, 0.3, 0.2, 2.0, interp), NULL, qq, tt);
      problem.SetParameterization(qq, new ceres::QuaternionParameterization);
      bad += check(problem.NumResidualBlocks() == 7 && g_deleted == 0, "blocks counted, nothing deleted while the problem lives");
    }
    bad += check(g_deleted == 3, "three distinct loss objects, each deleted once");
  }

  // the option / summary members the drivers touch
  Solver::Options options;
  options.linear_solver_type = ceres::DENSE_QR;
  options.minimizer_progress_to_stdout = false;
  options.max_num_iterations = 25;
  options.minimizer_type = ceres::TRUST_REGION;
  options.trust_region_strategy_type = ceres::DOGLEG;
  options.dogleg_type = ceres::TRADITIONAL_DOGLEG;
  Solver::Summary summary;
  LocalParameterization *par = new ceres::QuaternionParameterization;
  const double q[4] = {1, 0, 0, 0}, d[3] = {0.1, 0, 0};
  double qp[4], J[12];
  bad += check(par->Plus(q, d, qp) && std::fabs(qp[0] - std::cos(0.1)) < 1e-15 && std::fabs(qp[1] - std::sin(0.1)) < 1e-15, "Plus");
  bad += check(par->ComputeJacobian(q, J) && J[3] == 1.0 && J[7] == 1.0 && J[11] == 1.0 && J[0] == 0.0, "plus Jacobian at identity");
  bad += check(par->GlobalSize() == 4 && par->LocalSize() == 3, "sizes");
  delete par;
  std::string report = summary.FullReport();
  (void)report;
  std::printf("%s\n", bad ? "facade surface: FAILED" : "facade surface: ok");
  return bad;
}

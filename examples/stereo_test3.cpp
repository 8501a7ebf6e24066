// examples/stereo_test3.cpp — the solve block of the reference's stereo tests
// (ref: standalone/standalone_edge_align.cpp:778-815 for EAResidue + EAResidueSecondCam,
//  :3195-3233 for EAResidueEx + EAResidueSecondCamEx with TrivialLoss and 100 iterations,
//  :2590-2626 (tests 7-8) for EAResidue + EAResidueSecondCam with TrivialLoss, 100 iterations and every iterStep-th point,
//  iterStep = N / minNumOfPointsPerimage)
// compiled against the drop-in headers.  Input (binary, written by the Python tests):
//   int32 n1, n2, rows(H), cols(W); double K1[4], K2[4], Kc[5], trans_1to2[16], trans_1to2_inv[16];
//   double a_X[4*n1], a_X2[4*n2] (column-major 4xN); double e_disTrans[H*W], e_disTrans2[H*W] (column-major)
// argv[2] = "ex" selects the distortion flavour, "test7" the subsampled TrivialLoss flavour of tests 7-8.  Output: "q0 q1 q2 q3 t0 t1 t2 iterations termination".
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <vector>

#include "EAResidue.h"

using namespace ceres;

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const bool ex = argc > 2 && std::strcmp(argv[2], "ex") == 0;
  const bool test7 = argc > 2 && std::strcmp(argv[2], "test7") == 0;
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t hdr[4];
  double K1[4], K2[4], Kc[5], trans_1to2[16], trans_1to2_inv[16];
  if (std::fread(hdr, 4, 4, f) != 4 || std::fread(K1, 8, 4, f) != 4 || std::fread(K2, 8, 4, f) != 4 ||
      std::fread(Kc, 8, 5, f) != 5 || std::fread(trans_1to2, 8, 16, f) != 16 || std::fread(trans_1to2_inv, 8, 16, f) != 16)
    return 2;
  const int n1 = hdr[0], n2 = hdr[1], rows = hdr[2], cols = hdr[3];
  std::vector<double> a_X(4 * (size_t)n1), a_X2(4 * (size_t)n2), e_disTrans((size_t)rows * cols), e_disTrans2((size_t)rows * cols);
  if (std::fread(a_X.data(), 8, a_X.size(), f) != a_X.size() || std::fread(a_X2.data(), 8, a_X2.size(), f) != a_X2.size() ||
      std::fread(e_disTrans.data(), 8, e_disTrans.size(), f) != e_disTrans.size() ||
      std::fread(e_disTrans2.data(), 8, e_disTrans2.size(), f) != e_disTrans2.size())
    return 2;
  std::fclose(f);
  const double fx = K1[0], fy = K1[1], cx = K1[2], cy = K1[3], fx2 = K2[0], fy2 = K2[1], cx2 = K2[2], cy2 = K2[3];

  // ---- reference text (standalone_edge_align.cpp:778-815 / :3195-3233) ----------------------------
  ceres::Grid2D<double, 1> grid(e_disTrans.data(), 0, cols, 0, rows);
  ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interpolated_imb_disTrans(grid);
  ceres::Grid2D<double, 1> grid2(e_disTrans2.data(), 0, cols, 0, rows);
  ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interpolated_imb_disTrans2(grid2);

  double b_quat_a[10] = {1, 0, 0, 0}, b_t_a[10] = {0, 0, 0};
  ceres::Problem problem;
  int count = 0;
  const int minNumOfPointsPerimage = 1000;   // (:2590-2595)
  int iterStep = 1;
  if (test7 && n1 > minNumOfPointsPerimage) iterStep = n1 / minNumOfPointsPerimage;
  const bool trivial = ex || test7;
  for (int i = 0; i < n1; i += iterStep) {
    const double X = a_X[4 * (size_t)i], Y = a_X[4 * (size_t)i + 1], Z = a_X[4 * (size_t)i + 2];
    ceres::CostFunction *cost_function =
        ex ? EAResidueEx::Create(fx, fy, cx, cy, Kc[0], Kc[1], Kc[2], Kc[3], Kc[4], X, Y, Z, interpolated_imb_disTrans)
           : EAResidue::Create(fx, fy, cx, cy, X, Y, Z, interpolated_imb_disTrans);
    problem.AddResidualBlock(cost_function, trivial ? (LossFunction *)new TrivialLoss() : (LossFunction *)new CauchyLoss(1.), b_quat_a, b_t_a);
    count++;
  }
  for (int i = 0; i < n2; i += iterStep) {
    const double X = a_X2[4 * (size_t)i], Y = a_X2[4 * (size_t)i + 1], Z = a_X2[4 * (size_t)i + 2];
    ceres::CostFunction *cost_function =
        ex ? EAResidueSecondCamEx::Create(fx2, fy2, cx2, cy2, Kc[0], Kc[1], Kc[2], Kc[3], Kc[4], X, Y, Z, trans_1to2,
                                          trans_1to2_inv, interpolated_imb_disTrans2)
           : EAResidueSecondCam::Create(fx2, fy2, cx2, cy2, X, Y, Z, trans_1to2, trans_1to2_inv, interpolated_imb_disTrans2);
    problem.AddResidualBlock(cost_function, trivial ? (LossFunction *)new TrivialLoss() : (LossFunction *)new CauchyLoss(1.), b_quat_a, b_t_a);
    count++;
  }
  std::cerr << "-----> Use Point count = " << count << "\n";
  ceres::LocalParameterization *quaternion_parameterization = new ceres::QuaternionParameterization;
  problem.SetParameterization(b_quat_a, quaternion_parameterization);
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_QR;
  if (trivial) options.max_num_iterations = 100;
  Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);
  std::cerr << summary.BriefReport() << "\n";
  // ---- end of reference text ------------------------------------------------------------------------

  std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d\n", b_quat_a[0], b_quat_a[1], b_quat_a[2], b_quat_a[3],
              b_t_a[0], b_t_a[1], b_t_a[2], summary.num_successful_steps + summary.num_unsuccessful_steps,
              (int)summary.termination_type);
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

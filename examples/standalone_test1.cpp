// examples/standalone_test1.cpp — the solve block of edge_align_test1
// (ref: standalone/standalone_edge_align.cpp:256-300) compiled against the drop-in headers.
// Everything between the two "reference text" markers is the reference's call sequence with the
// Eigen containers replaced by plain arrays (Eigen/OpenCV are not in this image); inputs come
// from a small binary file written by the Python tests:
//   int32 N, int32 rows(H), int32 cols(W), double fx,fy,cx,cy, double a_X[4*N] (column-major 4xN),
//   double e_disTrans[H*W] (column-major H x W, as cv::cv2eigen fills an Eigen::MatrixXd)
// Output: one line "q0 q1 q2 q3 t0 t1 t2 iterations termination initial_cost final_cost".
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <vector>

#include "EAResidue.h"

using namespace ceres;

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s problem.bin [stride] [f32]\n", argv[0]); return 2; }
  const int stride = argc > 2 ? std::atoi(argv[2]) : 30;
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  int32_t N, rows, cols;
  double fx, fy, cx, cy;
  if (std::fread(&N, 4, 1, f) != 1 || std::fread(&rows, 4, 1, f) != 1 || std::fread(&cols, 4, 1, f) != 1) return 2;
  if (std::fread(&fx, 8, 1, f) != 1 || std::fread(&fy, 8, 1, f) != 1 || std::fread(&cx, 8, 1, f) != 1 || std::fread(&cy, 8, 1, f) != 1) return 2;
  std::vector<double> a_X(4 * (size_t)N), e_disTrans((size_t)rows * cols);
  if (std::fread(a_X.data(), 8, a_X.size(), f) != a_X.size()) return 2;
  if (std::fread(e_disTrans.data(), 8, e_disTrans.size(), f) != e_disTrans.size()) return 2;
  std::fclose(f);
  const int e_rows = rows, e_cols = cols;

  // ---- reference text (standalone_edge_align.cpp:256-293) ---------------------------------------
  ceres::Grid2D<double, 1> grid(e_disTrans.data(), 0, e_cols, 0, e_rows);
  ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interpolated_imb_disTrans(grid);

  double b_quat_a[10] = {1, 0, 0, 0}, b_t_a[10] = {0, 0, 0};  // eigenmat_to_raw(Identity)

  ceres::Problem problem;
  int count = 0;
  for (int i = 0; i < N; i += stride) {
    ceres::CostFunction *cost_function =
        EAResidue::Create(fx, fy, cx, cy, a_X[4 * (size_t)i + 0], a_X[4 * (size_t)i + 1], a_X[4 * (size_t)i + 2], interpolated_imb_disTrans);
    problem.AddResidualBlock(cost_function, new CauchyLoss(1.), b_quat_a, b_t_a);
    count++;
  }
  std::cerr << "Point count = " << count << "\n";

  ceres::LocalParameterization *quaternion_parameterization = new ceres::QuaternionParameterization;
  problem.SetParameterization(b_quat_a, quaternion_parameterization);

  // src/SolveEA.cpp:241-253: problem.Evaluate + the Num* queries, at the initial pose
  double cost0 = -1.0;
  std::vector<double> all_residues, grad0;
  const bool eval_ok = problem.Evaluate(ceres::Problem::EvaluateOptions(), &cost0, &all_residues, &grad0, NULL);
  double r2 = 0.0;
  for (double r : all_residues) r2 += r * r;
  // the same call with the Jacobian handed out (ceres::CRSMatrix, one dense 1x6 row per block): J^T r must be the gradient
  ceres::CRSMatrix jac0;
  std::vector<double> res_again;
  const bool jac_ok = problem.Evaluate(ceres::Problem::EvaluateOptions(), NULL, &res_again, NULL, &jac0);
  double jtr_err = 0.0, gmax = 0.0;
  if (jac_ok && jac0.num_rows == (int)res_again.size() && jac0.num_cols == 6 && grad0.size() == 6) {
    double g[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < jac0.num_rows; ++i)
      for (int k = jac0.rows[i]; k < jac0.rows[i + 1]; ++k) g[jac0.cols[k]] += jac0.values[k] * res_again[i];
    for (int a = 0; a < 6; ++a) { jtr_err = std::max(jtr_err, std::fabs(g[a] - grad0[a])); gmax = std::max(gmax, std::fabs(grad0[a])); }
  } else {
    jtr_err = 1e300;
  }
  std::cerr << "Evaluate: ok " << eval_ok << " cost " << cost0 << " residuals " << all_residues.size() << " NumParameterBlocks "
            << problem.NumParameterBlocks() << " NumParameters " << problem.NumParameters() << " NumResidualBlocks "
            << problem.NumResidualBlocks() << "\n";

  auto start1 = std::chrono::high_resolution_clock::now();
  ceres::Solver::Options options;
  options.minimizer_progress_to_stdout = false;
  options.linear_solver_type = ceres::DENSE_QR;
  if (argc > 3) options.ea_dtype = EA_F32;
  Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);
  auto finish1 = std::chrono::high_resolution_clock::now();
  std::chrono::duration<double> elapsed1 = finish1 - start1;
  std::cerr << "solve done. time = " << elapsed1.count() << "\n";
  std::cerr << summary.FullReport() << "\n";
  // ---- end of reference text --------------------------------------------------------------------

  std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d %.17g %.17g %d %.17g %d %.17g %d %d %.17g\n", b_quat_a[0], b_quat_a[1],
              b_quat_a[2], b_quat_a[3], b_t_a[0], b_t_a[1], b_t_a[2], summary.num_successful_steps + summary.num_unsuccessful_steps,
              (int)summary.termination_type, summary.initial_cost, summary.final_cost, (int)eval_ok, cost0,
              (int)all_residues.size(), 0.5 * r2, (int)grad0.size(), jac0.num_rows, gmax > 0 ? jtr_err / gmax : jtr_err);
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

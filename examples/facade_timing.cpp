// examples/facade_timing.cpp — what the drop-in costs per frame pair through the reference's own call sequence
// (standalone/standalone_edge_align.cpp:256-293): a ceres::Problem built block by block (EAResidue::Create +
// AddResidualBlock), SetParameterization, ceres::Solve, destruction -- repeated, so that the first repetition's one-time
// costs (GPU context, code object) are apart from the steady state.  Input file: see standalone_test1.cpp.
//   facade_timing problem.bin [stride] [repetitions]
// Prints per repetition: milliseconds for building the problem, for ceres::Solve, for destroying it; iterations.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "EAResidue.h"

using namespace ceres;

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s problem.bin [stride] [repetitions]\n", argv[0]); return 2; }
  const int stride = argc > 2 ? std::atoi(argv[2]) : 1, reps = argc > 3 ? std::atoi(argv[3]) : 8;
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
  using clk = std::chrono::steady_clock;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (int r = 0; r < reps; ++r) {
    const auto t0 = clk::now();
    ceres::Grid2D<double, 1> grid(e_disTrans.data(), 0, cols, 0, rows);
    ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interp(grid);
    double b_quat_a[10] = {1, 0, 0, 0}, b_t_a[10] = {0, 0, 0};
    int iters = 0, blocks = 0;
    double t_build, t_solve;
    clk::time_point t2;
    {
      ceres::Problem problem;
      for (int i = 0; i < N; i += stride) {
        ceres::CostFunction *cost_function =
            EAResidue::Create(fx, fy, cx, cy, a_X[4 * (size_t)i + 0], a_X[4 * (size_t)i + 1], a_X[4 * (size_t)i + 2], interp);
        problem.AddResidualBlock(cost_function, new CauchyLoss(1.), b_quat_a, b_t_a);
        ++blocks;
      }
      problem.SetParameterization(b_quat_a, new ceres::QuaternionParameterization);
      const auto t1 = clk::now();
      ceres::Solver::Options options;
      options.linear_solver_type = ceres::DENSE_QR;
      Solver::Summary summary;
      ceres::Solve(options, &problem, &summary);
      t2 = clk::now();
      t_build = ms(t0, t1); t_solve = ms(t1, t2);
      iters = summary.num_successful_steps + summary.num_unsuccessful_steps;
    }
    const auto t3 = clk::now();
    std::printf("rep %d blocks %d build %.3f ms solve %.3f ms destroy %.3f ms iterations %d q %.9f %.9f %.9f %.9f\n", r, blocks, t_build,
                t_solve, ms(t2, t3), iters, b_quat_a[0], b_quat_a[1], b_quat_a[2], b_quat_a[3]);
  }
  return 0;
}

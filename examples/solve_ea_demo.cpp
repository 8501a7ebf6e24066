// examples/solve_ea_demo.cpp — the call sequence of src/ea.cpp:184-191 against the drop-in SolveEA
// (OpenCV-free entry points; same binary input as standalone_test1, at half resolution the ROS
// node would use its own frames).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "ros/SolveEA.h"

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s problem.bin\n", argv[0]); return 2; }
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t N, rows, cols;
  double K4[4];
  if (std::fread(&N, 4, 1, f) != 1 || std::fread(&rows, 4, 1, f) != 1 || std::fread(&cols, 4, 1, f) != 1) return 2;
  if (std::fread(K4, 8, 4, f) != 4) return 2;
  std::vector<double> a_X(4 * (size_t)N), dt((size_t)rows * cols), pts;
  if (std::fread(a_X.data(), 8, a_X.size(), f) != a_X.size()) return 2;
  if (std::fread(dt.data(), 8, dt.size(), f) != dt.size()) return 2;
  std::fclose(f);
  for (int i = 0; i < N; i += 10) { pts.push_back(a_X[4 * (size_t)i]); pts.push_back(a_X[4 * (size_t)i + 1]); pts.push_back(a_X[4 * (size_t)i + 2]); }
  SolveEA ea;
  ea._sampleCERESProblem();
  ea.setRefPoints(pts.data(), (int)(pts.size() / 3));
  ea.setNowDistanceTransform(dt.data(), rows, cols);
  ea._verify3dPts();
  ea.setAsCERESProblem();
  double q[4], t[3];
  ea.getPose(q, t);
  std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n", q[0], q[1], q[2], q[3], t[0], t[1], t[2], (int)ea.summary().termination_type);
  return 0;
}

// The same probe for the ROS flavour's functor (include/EAResidue.h:69-126 upstream): constructor
// (lx, ly, lz, BiCubicInterpolator<Grid2D<double,2>>&, K), templated operator().  The grid on stdin is single-channel; it
// is duplicated into two interleaved channels here, as upstream's Grid2D<double,2> view expects.
#include <cstdio>
#include <vector>

#include "ros/EAResidue.h"

struct Mat3 {
  double m[9];
  double &operator()(int i, int j) { return m[3 * i + j]; }
};

int main() {
  int rows, cols;
  if (std::scanf("%d %d", &rows, &cols) != 2) return 2;
  std::vector<double> grid((size_t)rows * cols * 2);
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    if (std::scanf("%lf", &grid[2 * i]) != 1) return 2;
    grid[2 * i + 1] = -grid[2 * i];
  }
  double fx, fy, cx, cy;
  int n;
  if (std::scanf("%lf %lf %lf %lf %d", &fx, &fy, &cx, &cy, &n) != 5) return 2;
  Mat3 K = {{fx, 0, cx, 0, fy, cy, 0, 0, 1}};
  ceres::Grid2D<double, 2> g(grid.data(), 0, rows, 0, cols);
  ceres::BiCubicInterpolator<ceres::Grid2D<double, 2>> interp(g);
  typedef ceres::Jet<double, 7> J7;
  for (int i = 0; i < n; ++i) {
    double q[4], t[3], X[3];
    for (double &v : q) if (std::scanf("%lf", &v) != 1) return 2;
    for (double &v : t) if (std::scanf("%lf", &v) != 1) return 2;
    for (double &v : X) if (std::scanf("%lf", &v) != 1) return 2;
    EAResidue f(X[0], X[1], X[2], interp, K);
    double r = 0.0;
    const bool ok = f(q, t, &r);
    J7 jq[4], jt[3], jr;
    for (int k = 0; k < 4; ++k) jq[k] = J7(q[k], k);
    for (int k = 0; k < 3; ++k) jt[k] = J7(t[k], 4 + k);
    const bool okj = f(jq, jt, &jr);
    std::printf("%d %d %.17g %.17g", ok ? 1 : 0, okj ? 1 : 0, r, jr.a);
    for (int k = 0; k < 7; ++k) std::printf(" %.17g", jr.v[k]);
    std::printf("\n");
  }
  return 0;
}

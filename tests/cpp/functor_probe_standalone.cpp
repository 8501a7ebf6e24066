// The reference's templated call operator through the drop-in header (standalone flavour, utils.h:47-80):
// instantiated for double and for ceres::Jet<double, 7>.  Reads a problem from stdin, prints r and d r / d (q, t).
//   stdin: rows cols, rows*cols grid values, fx fy cx cy, n, then n lines: q(4) t(3) X(3)
#include <cstdio>
#include <vector>

#include "EAResidue.h"

int main() {
  int rows, cols;
  if (std::scanf("%d %d", &rows, &cols) != 2) return 2;
  std::vector<double> grid((size_t)rows * cols);
  for (double &g : grid)
    if (std::scanf("%lf", &g) != 1) return 2;
  double fx, fy, cx, cy;
  int n;
  if (std::scanf("%lf %lf %lf %lf %d", &fx, &fy, &cx, &cy, &n) != 5) return 2;
  ceres::Grid2D<double, 1> g(grid.data(), 0, rows, 0, cols);
  ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interp(g);
  typedef ceres::Jet<double, 7> J7;
  for (int i = 0; i < n; ++i) {
    double q[4], t[3], X[3];
    for (double &v : q) if (std::scanf("%lf", &v) != 1) return 2;
    for (double &v : t) if (std::scanf("%lf", &v) != 1) return 2;
    for (double &v : X) if (std::scanf("%lf", &v) != 1) return 2;
    EAResidue f(fx, fy, cx, cy, X[0], X[1], X[2], interp);
    double r = 0.0;
    const bool ok = f(q, t, &r);                       // operator()<double>
    J7 jq[4], jt[3], jr;
    for (int k = 0; k < 4; ++k) jq[k] = J7(q[k], k);
    for (int k = 0; k < 3; ++k) jt[k] = J7(t[k], 4 + k);
    const bool okj = f(jq, jt, &jr);                   // operator()<ceres::Jet<double, 7>>
    std::printf("%d %d %.17g %.17g", ok ? 1 : 0, okj ? 1 : 0, r, jr.a);
    for (int k = 0; k < 7; ++k) std::printf(" %.17g", jr.v[k]);
    std::printf("\n");
  }
  return 0;
}

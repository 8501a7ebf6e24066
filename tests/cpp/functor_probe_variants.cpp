// The three variant functors (EAResidueEx / EAResidueSecondCam / EAResidueSecondCamEx, utils.h:102-421) through the drop-in
// header, evaluated the way Ceres would: CostFunction::Evaluate of the AutoDiffCostFunction their Create() returns
// (Jet<double, 7> through the templated call operator).  Prints r, d r / d q (4), d r / d t (3) per case.
//   stdin: rows cols, grid, fx fy cx cy, variant (1 Ex, 2 SecondCam, 3 SecondCamEx), k1 k2 p1 p2 k3, T12 (16), T12inv (16),
//          n, then n lines: q(4) t(3) X(3)
#include <cstdio>
#include <memory>
#include <vector>

#include "EAResidue.h"

int main() {
  int rows, cols;
  if (std::scanf("%d %d", &rows, &cols) != 2) return 2;
  std::vector<double> grid((size_t)rows * cols);
  for (double &g : grid)
    if (std::scanf("%lf", &g) != 1) return 2;
  double fx, fy, cx, cy, k[5], T12[16], T12inv[16];
  int variant, n;
  if (std::scanf("%lf %lf %lf %lf %d", &fx, &fy, &cx, &cy, &variant) != 5) return 2;
  for (double &v : k) if (std::scanf("%lf", &v) != 1) return 2;
  for (double &v : T12) if (std::scanf("%lf", &v) != 1) return 2;
  for (double &v : T12inv) if (std::scanf("%lf", &v) != 1) return 2;
  if (std::scanf("%d", &n) != 1) return 2;
  ceres::Grid2D<double, 1> g(grid.data(), 0, rows, 0, cols);
  ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interp(g);
  for (int i = 0; i < n; ++i) {
    double q[4], t[3], X[3];
    for (double &v : q) if (std::scanf("%lf", &v) != 1) return 2;
    for (double &v : t) if (std::scanf("%lf", &v) != 1) return 2;
    for (double &v : X) if (std::scanf("%lf", &v) != 1) return 2;
    std::unique_ptr<ceres::CostFunction> cost(
        variant == 1   ? EAResidueEx::Create(fx, fy, cx, cy, k[0], k[1], k[2], k[3], k[4], X[0], X[1], X[2], interp)
        : variant == 2 ? EAResidueSecondCam::Create(fx, fy, cx, cy, X[0], X[1], X[2], T12, T12inv, interp)
        : variant == 3 ? EAResidueSecondCamEx::Create(fx, fy, cx, cy, k[0], k[1], k[2], k[3], k[4], X[0], X[1], X[2], T12, T12inv, interp)
                       : EAResidue::Create(fx, fy, cx, cy, X[0], X[1], X[2], interp));
    const double *params[2] = {q, t};
    double r = 0.0, r_only = 0.0, jq[4] = {0, 0, 0, 0}, jt[3] = {0, 0, 0};
    double *jac[2] = {jq, jt};
    const bool ok = cost->Evaluate(params, &r, jac);
    const bool ok2 = cost->Evaluate(params, &r_only, NULL);   // residual only: the <double> instantiation
    std::printf("%d %d %.17g %.17g", ok ? 1 : 0, ok2 ? 1 : 0, r_only, r);
    for (double v : jq) std::printf(" %.17g", v);
    for (double v : jt) std::printf(" %.17g", v);
    std::printf("\n");
  }
  return 0;
}

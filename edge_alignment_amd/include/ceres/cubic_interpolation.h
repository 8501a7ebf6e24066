// ceres/cubic_interpolation.h — Grid2D / BiCubicInterpolator carriers for the ceres:: facade.
// They record the view the reference builds (standalone_edge_align.cpp:258-259,
// src/SolveEA.cpp:152-154) so that ceres::Solve can hand it to ea_problem_set_dt.  The solver's
// sampling happens in the HIP kernels; Evaluate() below exists for the reference's host-side
// one-off probes (src/SolveEA.cpp:156-158) and follows the same Catmull-Rom definition.
#pragma once
#include <algorithm>
#include <cmath>

#include "jet.h"

namespace ceres {

template <typename T, int kDataDimension = 1, bool kRowMajor = true, bool kInterleaved = true>
class Grid2D {
 public:
  enum { DATA_DIMENSION = kDataDimension };
  Grid2D(const T *data, int row_begin, int row_end, int col_begin, int col_end)
      : data_(data), row_begin_(row_begin), row_end_(row_end), col_begin_(col_begin), col_end_(col_end),
        num_rows_(row_end - row_begin), num_cols_(col_end - col_begin) {}
  void GetValue(int r, int c, double *f) const {
    const int ri = std::min(std::max(row_begin_, r), row_end_ - 1) - row_begin_;
    const int ci = std::min(std::max(col_begin_, c), col_end_ - 1) - col_begin_;
    const int n = kRowMajor ? num_cols_ * ri + ci : num_rows_ * ci + ri;
    for (int i = 0; i < kDataDimension; ++i) f[i] = static_cast<double>(data_[kDataDimension * n + i]);
  }
  const T *data() const { return data_; }
  int num_rows() const { return num_rows_; }
  int num_cols() const { return num_cols_; }
  static constexpr bool row_major() { return kRowMajor; }

 private:
  const T *data_;
  int row_begin_, row_end_, col_begin_, col_end_, num_rows_, num_cols_;
};

template <typename Grid>
class BiCubicInterpolator {
 public:
  explicit BiCubicInterpolator(const Grid &grid) : grid_(grid) {}
  const Grid &grid() const { return grid_; }

  // host-side single sample (first data channel); not used by the solver
  void Evaluate(double r, double c, double *f, double *dfdr = nullptr, double *dfdc = nullptr) const {
    const int row = (int)std::floor(r), col = (int)std::floor(c);
    double fk[4], dk[4];
    for (int k = 0; k < 4; ++k) {
      double p[4][Grid::DATA_DIMENSION];
      for (int l = 0; l < 4; ++l) grid_.GetValue(row - 1 + k, col - 1 + l, p[l]);
      Spline(p[0][0], p[1][0], p[2][0], p[3][0], c - col, &fk[k], &dk[k]);
    }
    Spline(fk[0], fk[1], fk[2], fk[3], r - row, f, dfdr);
    if (dfdc) Spline(dk[0], dk[1], dk[2], dk[3], r - row, dfdc, nullptr);
  }

  // the overload Ceres provides for automatic differentiation: value from the scalar parts, derivatives by the chain
  // rule through dfdr, dfdc (ceres/cubic_interpolation.h, BiCubicInterpolator::Evaluate(const JetT&, const JetT&, JetT*))
  template <typename T, int N>
  void Evaluate(const Jet<T, N> &r, const Jet<T, N> &c, Jet<T, N> *f) const {
    double frc, dfdr, dfdc;
    Evaluate(static_cast<double>(r.a), static_cast<double>(c.a), &frc, &dfdr, &dfdc);
    f->a = T(frc);
    for (int i = 0; i < N; ++i) f->v[i] = T(dfdr) * r.v[i] + T(dfdc) * c.v[i];
  }

 private:
  static void Spline(double p0, double p1, double p2, double p3, double x, double *f, double *dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    if (f) *f = p1 + x * (c + x * (b + x * a));
    if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
  }
  const Grid &grid_;
};

}  // namespace ceres

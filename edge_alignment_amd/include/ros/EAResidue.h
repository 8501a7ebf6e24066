// ros/EAResidue.h — drop-in for include/EAResidue.h:69-126 of kuwt/edge_alignment (ROS flavour).
//
// Keeps the constructor `EAResidue(lx,ly,lz, BiCubicInterpolator<Grid2D<double,2>>&, Matrix3d& K)`.
// K is any type indexable as K(i,j) (Eigen::Matrix3d upstream).  The upstream functor is
// numerically broken (it reads a single-channel image through a 2-channel grid
// and writes two values into a 1-slot residual); this drop-in keeps its *intended* semantics as
// kernel knobs — R applied transposed (:99-101), divisor z + 0.001 (:104-105), no z guard — and
// samples channel 0 of the grid it is given.  It is an API surface, not a parity target.
#pragma once
#include "../ceres/ceres.h"

class EAResidue {
 public:
  typedef ceres::BiCubicInterpolator<ceres::Grid2D<double, 2>> Interpolator2;
  typedef ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> Interpolator1;

  template <typename Mat3>
  EAResidue(double lx, double ly, double lz, Interpolator2 &interpolated_a, Mat3 &K)
      : lx(lx), ly(ly), lz(lz), data_(interpolated_a.grid().data()), rows_(interpolated_a.grid().num_rows()),
        cols_(interpolated_a.grid().num_cols()), channels_(2) {
    fx = K(0, 0); fy = K(1, 1); cx = K(0, 2); cy = K(1, 2);
  }
  template <typename Mat3>
  EAResidue(double lx, double ly, double lz, Interpolator1 &interpolated_a, Mat3 &K)
      : lx(lx), ly(ly), lz(lz), data_(interpolated_a.grid().data()), rows_(interpolated_a.grid().num_rows()),
        cols_(interpolated_a.grid().num_cols()), channels_(1) {
    fx = K(0, 0); fy = K(1, 1); cx = K(0, 2); cy = K(1, 2);
  }

  // The reference's call operator, `template <typename T>` as upstream (include/EAResidue.h:85-118):
  // ceres::QuaternionToRotation, the row-major result read column-wise (:99-101, i.e. R transposed), divisor
  // z + 0.001 (:104-105), no guard.  Channel 0 of the grid is sampled (upstream writes both channels of its
  // Grid2D<double,2> into the one-slot residual: SURVEY App. D).  Host-side probe; the solver never calls it.
  template <typename T>
  bool operator()(const T *const Q, const T *const t, T *residual) const {
    T R[9];
    ceres::QuaternionToRotation(Q, R);
    const T _x = T(lx), _y = T(ly), _z = T(lz);
    const T _xd = t[0] + R[0] * _x + R[3] * _y + R[6] * _z;
    const T _yd = t[1] + R[1] * _x + R[4] * _y + R[7] * _z;
    const T _zd = t[2] + R[2] * _x + R[5] * _y + R[8] * _z;
    const T _u = T(fx) * _xd / (_zd + T(.001)) + T(cx);
    const T _v = T(fy) * _yd / (_zd + T(.001)) + T(cy);
    if (channels_ == 2) {
      const ceres::Grid2D<double, 2> g(data_, 0, rows_, 0, cols_);
      ceres::BiCubicInterpolator<ceres::Grid2D<double, 2>>(g).Evaluate(_u, _v, &residual[0]);
    } else {
      const ceres::Grid2D<double, 1> g(data_, 0, rows_, 0, cols_);
      ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>>(g).Evaluate(_u, _v, &residual[0]);
    }
    return true;
  }

  bool ea_describe(ceres::EABlockInfo *b) const {
    b->fx = fx; b->fy = fy; b->cx = cx; b->cy = cy;
    b->X = lx; b->Y = ly; b->Z = lz;
    b->grid_data = data_; b->grid_rows = rows_; b->grid_cols = cols_;
    b->z_guard = 0.0; b->z_eps = 0.001; b->rot_transposed = 1;
    return true;
  }

 private:
  double lx, ly, lz;
  double fx, fy, cx, cy;
  const double *data_;
  int rows_, cols_, channels_;
};

// EAResidue.h — drop-in for the cost functor of standalone/utils.h:38-99 (kuwt/edge_alignment).
//
// Same class name, constructor and static Create(...) as the reference, so the loop at
// standalone/standalone_edge_align.cpp:267-274 compiles unchanged:
//     ceres::CostFunction* cost = EAResidue::Create(fx,fy,cx,cy, X,Y,Z, interpolated_imb_disTrans);
//     problem.AddResidualBlock(cost, new CauchyLoss(1.), b_quat_a, b_t_a);
// The functor no longer computes anything per call through dual numbers: it carries the block's
// constants, and ceres::Solve (facade, ceres/ceres.h) evaluates all blocks at once in the gfx950
// kernels behind include/ea_hip.h.  The templated operator() is kept, instantiable for double and for ceres::Jet, for
// host-side spot checks; it follows the reference text line by line (utils.h:48-80).
#pragma once
#include "ceres/ceres.h"

class EAResidue {
 public:
  typedef ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> Interpolator;

  EAResidue(const double fx, const double fy, const double cx, const double cy, const double a_Xx,
            const double a_Xy, const double a_Xz, const Interpolator &__interpolated_a)
      : interp_a(__interpolated_a), fx(fx), fy(fy), cx(cx), cy(cy), a_Xx(a_Xx), a_Xy(a_Xy), a_Xz(a_Xz) {}

  // The reference's call operator, `template <typename T>` as upstream (utils.h:47-80): b = R(q) a + t with Eigen's
  // un-normalised toRotationMatrix formula written out (:51-53), the |b_z| < 0.01 guard (:70-73), pinhole (:74-75), the
  // interpolator (:77).  T = double samples the DT on the host; T = ceres::Jet<double, 7> also yields d r / d (q, t).
  // A host-side probe: ceres::Solve never calls it, every block is evaluated by the gfx950 kernels.
  template <typename T>
  bool operator()(const T *const quat, const T *const t, T *residue) const {
    const T w = quat[0], x = quat[1], y = quat[2], z = quat[3];
    const T one(1.0), two(2.0);
    const T R[9] = {one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y),
                    two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x),
                    two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)};
    const T bx = R[0] * T(a_Xx) + R[1] * T(a_Xy) + R[2] * T(a_Xz) + t[0];
    const T by = R[3] * T(a_Xx) + R[4] * T(a_Xy) + R[5] * T(a_Xz) + t[1];
    const T bz = R[6] * T(a_Xx) + R[7] * T(a_Xy) + R[8] * T(a_Xz) + t[2];
    if (bz < T(0.01) && bz > T(-0.01)) return false;
    const T _u = T(fx) * bx / bz + T(cx);
    const T _v = T(fy) * by / bz + T(cy);
    interp_a.Evaluate(_u, _v, &residue[0]);
    return true;
  }

  static ceres::CostFunction *Create(const double fx, const double fy, const double cx, const double cy,
                                     const double a_Xx, const double a_Xy, const double a_Xz,
                                     const Interpolator &__interpolated_a) {
    return (new ceres::AutoDiffCostFunction<EAResidue, 1, 4, 3>(
        new EAResidue(fx, fy, cx, cy, a_Xx, a_Xy, a_Xz, __interpolated_a)));
  }

  // read by the ceres:: facade when the problem is handed to the GPU
  bool ea_describe(ceres::EABlockInfo *b) const {
    b->fx = fx; b->fy = fy; b->cx = cx; b->cy = cy;
    b->X = a_Xx; b->Y = a_Xy; b->Z = a_Xz;
    b->grid_data = interp_a.grid().data();
    b->grid_rows = interp_a.grid().num_rows();
    b->grid_cols = interp_a.grid().num_cols();
    b->z_guard = 0.01; b->z_eps = 0.0; b->rot_transposed = 0;  // utils.h:70-75
    return true;
  }

 private:
  const Interpolator &interp_a;
  double fx, fy, cx, cy;
  double a_Xx, a_Xy, a_Xz;
};

// ---- variants of standalone/utils.h:102-421: carriers with the reference's constructors and Create() -----

namespace ea_shim {
// Host-side evaluation shared by the three variant functors' templated call operators (T = double or ceres::Jet):
//   a' = T12inv a;  b' = R(q) a' + t;  b = T12 b'   (i.e. b = (T12 * b_T_a * T12inv) a, utils.h:244-259; rig == NULL: b = b')
//   |b_z| < 0.01 -> false (:262-265);  x = b_x / b_z, y = b_y / b_z;  Brown-Conrady (k1, k2, p1, p2, k3) when dist != NULL
//   (:137-143);  u = fx x_d + cx, v = fy y_d + cy;  residue = interpolator(u, v).
// The solver never calls this: ceres::Solve evaluates every block in the gfx950 kernels (project_point_var).
template <typename T, typename Interp>
bool variant_residue(const T *const quat, const T *const t, T *residue, const Interp &interp, double fx, double fy, double cx,
                     double cy, double X, double Y, double Z, const double *dist, const double *T12, const double *T12inv) {
  double a[3] = {X, Y, Z};
  if (T12inv) {
    for (int i = 0; i < 3; ++i) a[i] = T12inv[4 * i] * X + T12inv[4 * i + 1] * Y + T12inv[4 * i + 2] * Z + T12inv[4 * i + 3];
  }
  const T w = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
  const T one(1.0), two(2.0);
  const T R[9] = {one - two * (qy * qy + qz * qz), two * (qx * qy - w * qz), two * (qx * qz + w * qy),
                  two * (qx * qy + w * qz), one - two * (qx * qx + qz * qz), two * (qy * qz - w * qx),
                  two * (qx * qz - w * qy), two * (qy * qz + w * qx), one - two * (qx * qx + qy * qy)};
  T b[3];
  for (int i = 0; i < 3; ++i) b[i] = R[3 * i] * T(a[0]) + R[3 * i + 1] * T(a[1]) + R[3 * i + 2] * T(a[2]) + t[i];
  if (T12) {
    T c[3];
    for (int i = 0; i < 3; ++i) c[i] = T(T12[4 * i]) * b[0] + T(T12[4 * i + 1]) * b[1] + T(T12[4 * i + 2]) * b[2] + T(T12[4 * i + 3]);
    for (int i = 0; i < 3; ++i) b[i] = c[i];
  }
  if (b[2] < T(0.01) && b[2] > T(-0.01)) return false;
  const T x = b[0] / b[2], y = b[1] / b[2];
  T xd = x, yd = y;
  if (dist) {
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    const T r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const T radial = T(1.0) + T(k1) * r2 + T(k2) * r4 + T(k3) * r6;
    xd = x * radial + T(2.0 * p1) * x * y + T(p2) * (r2 + T(2.0) * x * x);
    yd = y * radial + T(2.0 * p2) * x * y + T(p1) * (r2 + T(2.0) * y * y);
  }
  const T u = T(fx) * xd + T(cx);
  const T v = T(fy) * yd + T(cy);
  interp.Evaluate(u, v, &residue[0]);
  return true;
}
}  // namespace ea_shim

// distortion (utils.h:102-177)
class EAResidueEx {
 public:
  typedef EAResidue::Interpolator Interpolator;
  EAResidueEx(const double fx, const double fy, const double cx, const double cy, const double k1, const double k2,
              const double p1, const double p2, const double k3, const double a_Xx, const double a_Xy,
              const double a_Xz, const Interpolator &__interpolated_a)
      : interp_a(__interpolated_a), fx(fx), fy(fy), cx(cx), cy(cy), k1(k1), k2(k2), p1(p1), p2(p2), k3(k3),
        a_Xx(a_Xx), a_Xy(a_Xy), a_Xz(a_Xz) {}
  // `template <typename T>` as upstream; host-side probe (see ea_shim::variant_residue)
  template <typename T>
  bool operator()(const T *const quat, const T *const t, T *residue) const {
    const double d[5] = {k1, k2, p1, p2, k3};
    return ea_shim::variant_residue(quat, t, residue, interp_a, fx, fy, cx, cy, a_Xx, a_Xy, a_Xz, d, nullptr, nullptr);
  }
  static ceres::CostFunction *Create(const double fx, const double fy, const double cx, const double cy,
                                     const double k1, const double k2, const double p1, const double p2,
                                     const double k3, const double a_Xx, const double a_Xy, const double a_Xz,
                                     const Interpolator &__interpolated_a) {
    return (new ceres::AutoDiffCostFunction<EAResidueEx, 1, 4, 3>(
        new EAResidueEx(fx, fy, cx, cy, k1, k2, p1, p2, k3, a_Xx, a_Xy, a_Xz, __interpolated_a)));
  }
  bool ea_describe(ceres::EABlockInfo *b) const {
    b->fx = fx; b->fy = fy; b->cx = cx; b->cy = cy;
    b->X = a_Xx; b->Y = a_Xy; b->Z = a_Xz;
    b->grid_data = interp_a.grid().data(); b->grid_rows = interp_a.grid().num_rows(); b->grid_cols = interp_a.grid().num_cols();
    b->z_guard = 0.01; b->z_eps = 0.0; b->rot_transposed = 0;
    b->variant = 1;
    b->dist[0] = k1; b->dist[1] = k2; b->dist[2] = p1; b->dist[3] = p2; b->dist[4] = k3;
    return true;
  }

 private:
  const Interpolator &interp_a;
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
  double a_Xx, a_Xy, a_Xz;
};

// second camera of a rigid rig (utils.h:179-292): b_T_a_SecCam = trans_1to2 * b_T_a * trans_1to2_inv
class EAResidueSecondCam {
 public:
  typedef EAResidue::Interpolator Interpolator;
  EAResidueSecondCam(const double fx, const double fy, const double cx, const double cy, const double a_Xx,
                     const double a_Xy, const double a_Xz, const double *ptrans_1to2, const double *ptrans_1to2_inv,
                     const Interpolator &__interpolated_a)
      : interp_a(__interpolated_a), fx(fx), fy(fy), cx(cx), cy(cy), a_Xx(a_Xx), a_Xy(a_Xy), a_Xz(a_Xz) {
    for (int i = 0; i < 16; ++i) { T12[i] = ptrans_1to2[i]; T12inv[i] = ptrans_1to2_inv[i]; }
  }
  // `template <typename T>` as upstream; host-side probe (see ea_shim::variant_residue)
  template <typename T>
  bool operator()(const T *const quat, const T *const t, T *residue) const {
    return ea_shim::variant_residue(quat, t, residue, interp_a, fx, fy, cx, cy, a_Xx, a_Xy, a_Xz, nullptr, T12, T12inv);
  }
  static ceres::CostFunction *Create(const double fx, const double fy, const double cx, const double cy,
                                     const double a_Xx, const double a_Xy, const double a_Xz,
                                     const double *ptrans_1to2, const double *ptrans_1to2_inv,
                                     const Interpolator &__interpolated_a) {
    return (new ceres::AutoDiffCostFunction<EAResidueSecondCam, 1, 4, 3>(
        new EAResidueSecondCam(fx, fy, cx, cy, a_Xx, a_Xy, a_Xz, ptrans_1to2, ptrans_1to2_inv, __interpolated_a)));
  }
  bool ea_describe(ceres::EABlockInfo *b) const {
    b->fx = fx; b->fy = fy; b->cx = cx; b->cy = cy;
    b->X = a_Xx; b->Y = a_Xy; b->Z = a_Xz;
    b->grid_data = interp_a.grid().data(); b->grid_rows = interp_a.grid().num_rows(); b->grid_cols = interp_a.grid().num_cols();
    b->z_guard = 0.01; b->z_eps = 0.0; b->rot_transposed = 0;
    b->variant = 2;
    for (int i = 0; i < 16; ++i) { b->T12[i] = T12[i]; b->T12inv[i] = T12inv[i]; }
    return true;
  }

 private:
  const Interpolator &interp_a;
  double fx, fy, cx, cy;
  double a_Xx, a_Xy, a_Xz;
  double T12[16], T12inv[16];
};

// second camera + distortion (utils.h:295-421)
class EAResidueSecondCamEx {
 public:
  typedef EAResidue::Interpolator Interpolator;
  EAResidueSecondCamEx(const double fx, const double fy, const double cx, const double cy, const double k1,
                       const double k2, const double p1, const double p2, const double k3, const double a_Xx,
                       const double a_Xy, const double a_Xz, const double *ptrans_1to2,
                       const double *ptrans_1to2_inv, const Interpolator &__interpolated_a)
      : interp_a(__interpolated_a), fx(fx), fy(fy), cx(cx), cy(cy), k1(k1), k2(k2), p1(p1), p2(p2), k3(k3),
        a_Xx(a_Xx), a_Xy(a_Xy), a_Xz(a_Xz) {
    for (int i = 0; i < 16; ++i) { T12[i] = ptrans_1to2[i]; T12inv[i] = ptrans_1to2_inv[i]; }
  }
  // `template <typename T>` as upstream; host-side probe (see ea_shim::variant_residue)
  template <typename T>
  bool operator()(const T *const quat, const T *const t, T *residue) const {
    const double d[5] = {k1, k2, p1, p2, k3};
    return ea_shim::variant_residue(quat, t, residue, interp_a, fx, fy, cx, cy, a_Xx, a_Xy, a_Xz, d, T12, T12inv);
  }
  static ceres::CostFunction *Create(const double fx, const double fy, const double cx, const double cy,
                                     const double k1, const double k2, const double p1, const double p2,
                                     const double k3, const double a_Xx, const double a_Xy, const double a_Xz,
                                     const double *ptrans_1to2, const double *ptrans_1to2_inv,
                                     const Interpolator &__interpolated_a) {
    return (new ceres::AutoDiffCostFunction<EAResidueSecondCamEx, 1, 4, 3>(new EAResidueSecondCamEx(
        fx, fy, cx, cy, k1, k2, p1, p2, k3, a_Xx, a_Xy, a_Xz, ptrans_1to2, ptrans_1to2_inv, __interpolated_a)));
  }
  bool ea_describe(ceres::EABlockInfo *b) const {
    b->fx = fx; b->fy = fy; b->cx = cx; b->cy = cy;
    b->X = a_Xx; b->Y = a_Xy; b->Z = a_Xz;
    b->grid_data = interp_a.grid().data(); b->grid_rows = interp_a.grid().num_rows(); b->grid_cols = interp_a.grid().num_cols();
    b->z_guard = 0.01; b->z_eps = 0.0; b->rot_transposed = 0;
    b->variant = 3;
    b->dist[0] = k1; b->dist[1] = k2; b->dist[2] = p1; b->dist[3] = p2; b->dist[4] = k3;
    for (int i = 0; i < 16; ++i) { b->T12[i] = T12[i]; b->T12inv[i] = T12inv[i]; }
    return true;
  }

 private:
  const Interpolator &interp_a;
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
  double a_Xx, a_Xy, a_Xz;
  double T12[16], T12inv[16];
};

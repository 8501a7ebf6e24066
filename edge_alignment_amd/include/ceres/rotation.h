// ceres/rotation.h — QuaternionToRotation as ceres/rotation.h defines it (row-major 3x3 from q = (w, x, y, z), scaled
// by 1 / |q|^2), the one rotation helper the reference calls (include/EAResidue.h:90).  Templated: works for double and
// for ceres::Jet.
#pragma once

namespace ceres {

template <typename T>
inline void QuaternionToScaledRotation(const T q[4], T R[3 * 3]) {
  const T a = q[0], b = q[1], c = q[2], d = q[3];
  const T aa = a * a, ab = a * b, ac = a * c, ad = a * d, bb = b * b, bc = b * c, bd = b * d, cc = c * c, cd = c * d,
          dd = d * d;
  R[0] = aa + bb - cc - dd; R[1] = T(2) * (bc - ad);  R[2] = T(2) * (ac + bd);
  R[3] = T(2) * (ad + bc);  R[4] = aa - bb + cc - dd; R[5] = T(2) * (cd - ab);
  R[6] = T(2) * (bd - ac);  R[7] = T(2) * (ab + cd);  R[8] = aa - bb - cc + dd;
}

template <typename T>
inline void QuaternionToRotation(const T q[4], T R[3 * 3]) {
  QuaternionToScaledRotation(q, R);
  T normalizer = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  normalizer = T(1) / normalizer;
  for (int i = 0; i < 9; ++i) R[i] = R[i] * normalizer;
}

}  // namespace ceres

/*
 * ea_oracle.c — see ea_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no Ceres here).
 *
 * Every function cites what it follows.  "ref:" = file under /root/reference (kuwt/edge_alignment),
 * "ceres:" = the published Ceres Solver (<= 2.1) algorithm, restated from its public description
 * because the library is absent from the reference tree and from this image.
 *
 * Compile with -ffp-contract=off so the arithmetic is plain IEEE double, one rounding per op.
 */
#include "ea_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* defaults: ceres::Solver::Options as test1 leaves them (ref: standalone_edge_align.cpp:282-284) */

void ea_oracle_default_options(ea_oracle_options *o) {
  o->max_num_iterations = 50;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->jacobian_mode = EA_ORACLE_JAC_ANALYTIC;
  o->linear_solver = EA_ORACLE_LIN_CHOLESKY;
  o->strategy = EA_ORACLE_STRATEGY_LM;
  o->verbose = 0;
}

void ea_oracle_default_problem(ea_oracle_problem *p) {
  memset(p, 0, sizeof(*p));
  p->loss_kind = EA_ORACLE_LOSS_CAUCHY; /* ref: standalone_edge_align.cpp:272 */
  p->loss_a = 1.0;
  p->z_guard = 0.01;                    /* ref: standalone/utils.h:70 */
  p->z_eps = 0.0;
  p->rot_transposed = 0;
  p->use_distortion = 0;
  p->use_second_cam = 0;
  for (int i = 0; i < 16; ++i) p->T12[i] = p->T12inv[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* ceres: Grid2D<double,1>::GetValue — clamp-to-edge, row-major                                */

static inline double grid_value(const double *grid, int rows, int cols, int r, int c) {
  int ri = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r);
  int ci = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
  return grid[(size_t)cols * (size_t)ri + (size_t)ci];
}

/* ceres: CubicHermiteSpline<1>(p0,p1,p2,p3,x,f,dfdx) — Catmull-Rom, Horner form */
static inline void cubic_hermite(double p0, double p1, double p2, double p3, double x, double *f,
                                 double *dfdx) {
  const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
  const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
  const double c = 0.5 * (-p0 + p2);
  const double d = p1;
  if (f) *f = d + x * (c + x * (b + x * a));
  if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

/* floor to int with saturation (Ceres does `const int row = std::floor(r)`; far-outside
 * coordinates all clamp to the border texel anyway, so saturating keeps the same value) */
static inline int floor_sat(double x) {
  double f = floor(x);
  if (!(f > -1.0e9)) return -1000000000; /* also catches NaN */
  if (f > 1.0e9) return 1000000000;
  return (int)f;
}

/* ceres: BiCubicInterpolator<Grid>::Evaluate(r, c, f, dfdr, dfdc)
 * Four splines along the columns (one per grid row row-1..row+2), then one spline across rows
 * for the value and one for d/dc. */
void ea_oracle_bicubic(const double *grid, int rows, int cols, double r, double c, double *f,
                       double *dfdr, double *dfdc) {
  const int row = floor_sat(r);
  const int col = floor_sat(c);
  double fk[4], dfk[4];
  for (int k = 0; k < 4; ++k) {
    const double p0 = grid_value(grid, rows, cols, row - 1 + k, col - 1);
    const double p1 = grid_value(grid, rows, cols, row - 1 + k, col);
    const double p2 = grid_value(grid, rows, cols, row - 1 + k, col + 1);
    const double p3 = grid_value(grid, rows, cols, row - 1 + k, col + 2);
    cubic_hermite(p0, p1, p2, p3, c - col, &fk[k], &dfk[k]);
  }
  cubic_hermite(fk[0], fk[1], fk[2], fk[3], r - row, f, dfdr);
  if (dfdc) cubic_hermite(dfk[0], dfk[1], dfk[2], dfk[3], r - row, dfdc, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* ceres: Jet<double,7> (jet.h) — just the operators the functor touches                       */

#define NJ 7
typedef struct {
  double a;
  double v[NJ];
} jet;

static jet jet_const(double a) {
  jet j;
  j.a = a;
  for (int i = 0; i < NJ; ++i) j.v[i] = 0.0;
  return j;
}
static jet jet_var(double a, int k) {
  jet j = jet_const(a);
  j.v[k] = 1.0;
  return j;
}
static jet jet_add(jet f, jet g) {
  jet h;
  h.a = f.a + g.a;
  for (int i = 0; i < NJ; ++i) h.v[i] = f.v[i] + g.v[i];
  return h;
}
static jet jet_sub(jet f, jet g) {
  jet h;
  h.a = f.a - g.a;
  for (int i = 0; i < NJ; ++i) h.v[i] = f.v[i] - g.v[i];
  return h;
}
/* ceres: Jet(f.a * g.a, f.a * g.v + f.v * g.a) */
static jet jet_mul(jet f, jet g) {
  jet h;
  h.a = f.a * g.a;
  for (int i = 0; i < NJ; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return h;
}
/* ceres: g_a_inverse = 1/g.a; f_a_by_g_a = f.a*g_a_inverse; Jet(f_a_by_g_a, (f.v - f_a_by_g_a*g.v)*g_a_inverse) */
static jet jet_div(jet f, jet g) {
  jet h;
  const double g_a_inverse = 1.0 / g.a;
  const double f_a_by_g_a = f.a * g_a_inverse;
  h.a = f_a_by_g_a;
  for (int i = 0; i < NJ; ++i) h.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return h;
}
static jet jet_scale(double s, jet f) { /* Eigen's Scalar(2)*x etc. become Jet*Jet with a constant Jet */
  return jet_mul(jet_const(s), f);
}

/* ref: standalone/utils.h:48-80  EAResidue::operator()<T> with T = Jet<double,7>
 * (include/EAResidue.h:86-118 differs by the knobs in ea_oracle_problem) */
static int functor_jet(const ea_oracle_problem *p, const jet quat[4], const jet t[3],
                       const double X[3], jet *residue) {
  jet R[3][3];
  const jet w = quat[0], x = quat[1], y = quat[2], z = quat[3];
  if (!p->rot_transposed || 1) {
    /* Eigen::QuaternionBase::toRotationMatrix (no normalisation) — utils.h:51-53 */
    const jet tx = jet_scale(2.0, x), ty = jet_scale(2.0, y), tz = jet_scale(2.0, z);
    const jet twx = jet_mul(tx, w), twy = jet_mul(ty, w), twz = jet_mul(tz, w);
    const jet txx = jet_mul(tx, x), txy = jet_mul(ty, x), txz = jet_mul(tz, x);
    const jet tyy = jet_mul(ty, y), tyz = jet_mul(tz, y), tzz = jet_mul(tz, z);
    const jet one = jet_const(1.0);
    R[0][0] = jet_sub(one, jet_add(tyy, tzz));
    R[0][1] = jet_sub(txy, twz);
    R[0][2] = jet_add(txz, twy);
    R[1][0] = jet_add(txy, twz);
    R[1][1] = jet_sub(one, jet_add(txx, tzz));
    R[1][2] = jet_sub(tyz, twx);
    R[2][0] = jet_sub(txz, twy);
    R[2][1] = jet_add(tyz, twx);
    R[2][2] = jet_sub(one, jet_add(txx, tyy));
  }
  if (p->rot_transposed) { /* include/EAResidue.h:99-101 indexes the row-major R column-wise */
    for (int i = 0; i < 3; ++i)
      for (int j = i + 1; j < 3; ++j) {
        jet tmp = R[i][j];
        R[i][j] = R[j][i];
        R[j][i] = tmp;
      }
  }
  jet b[3];
  if (!p->use_second_cam) {
    /* b_X = b_T_a * [a;1]  — utils.h:54-67 ; row i: R(i,0)a0 + R(i,1)a1 + R(i,2)a2 + t(i)*1 */
    for (int i = 0; i < 3; ++i) {
      jet s = jet_mul(R[i][0], jet_const(X[0]));
      s = jet_add(s, jet_mul(R[i][1], jet_const(X[1])));
      s = jet_add(s, jet_mul(R[i][2], jet_const(X[2])));
      s = jet_add(s, jet_mul(t[i], jet_const(1.0)));
      b[i] = s;
    }
  } else {
    /* utils.h:242-256: b_T_a_SecCam = TransformFromFirstCam * b_T_a * TransformFromFirstCamInv  (4x4 Jet
     * products, left to right), then b_X = b_T_a_SecCam * [a;1] */
    jet M[4][4], L[4][4], S[4][4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) M[i][j] = jet_const(0.0);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[i][j] = R[i][j];
      M[i][3] = t[i];
    }
    M[3][3] = jet_const(1.0);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        jet acc = jet_const(0.0);
        for (int k = 0; k < 4; ++k) acc = jet_add(acc, jet_mul(jet_const(p->T12[4 * i + k]), M[k][j]));
        L[i][j] = acc;
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        jet acc = jet_const(0.0);
        for (int k = 0; k < 4; ++k) acc = jet_add(acc, jet_mul(L[i][k], jet_const(p->T12inv[4 * k + j])));
        S[i][j] = acc;
      }
    for (int i = 0; i < 3; ++i) {
      jet s = jet_mul(S[i][0], jet_const(X[0]));
      s = jet_add(s, jet_mul(S[i][1], jet_const(X[1])));
      s = jet_add(s, jet_mul(S[i][2], jet_const(X[2])));
      s = jet_add(s, jet_mul(S[i][3], jet_const(1.0)));
      b[i] = s;
    }
  }
  /* z guard — utils.h:70-73 (comparisons look at the scalar part only) */
  if (p->z_guard > 0.0 && b[2].a < p->z_guard && b[2].a > -p->z_guard) return 0;
  jet bz = b[2];
  if (p->z_eps != 0.0) bz = jet_add(bz, jet_const(p->z_eps));
  jet u, v;
  if (!p->use_distortion) {
    /* _u = T(fx)*b_X(0)/b_X(2) + T(cx) — utils.h:74-75 */
    u = jet_add(jet_div(jet_mul(jet_const(p->fx), b[0]), bz), jet_const(p->cx));
    v = jet_add(jet_div(jet_mul(jet_const(p->fy), b[1]), bz), jet_const(p->cy));
  } else {
    /* utils.h:140-149 (Brown-Conrady): x = X/Z, y = Y/Z, r2, r4, r6,
     * distort_x = x (1 + k1 r2 + k2 r4 + k3 r6) + 2 p1 x y + p2 (r2 + 2 x x), distort_y likewise */
    const jet x = jet_div(b[0], bz), y = jet_div(b[1], bz);
    const jet r2 = jet_add(jet_mul(x, x), jet_mul(y, y));
    const jet r4 = jet_mul(r2, r2), r6 = jet_mul(r4, r2);
    const jet radial = jet_add(jet_add(jet_add(jet_const(1.0), jet_scale(p->k1, r2)), jet_scale(p->k2, r4)), jet_scale(p->k3, r6));
    const jet dx = jet_add(jet_add(jet_mul(x, radial), jet_mul(jet_mul(jet_const(2.0 * p->p1), x), y)),
                           jet_scale(p->p2, jet_add(r2, jet_mul(jet_scale(2.0, x), x))));
    const jet dy = jet_add(jet_add(jet_mul(y, radial), jet_mul(jet_mul(jet_const(2.0 * p->p2), x), y)),
                           jet_scale(p->p1, jet_add(r2, jet_mul(jet_scale(2.0, y), y))));
    u = jet_add(jet_mul(jet_const(p->fx), dx), jet_const(p->cx));
    v = jet_add(jet_mul(jet_const(p->fy), dy), jet_const(p->cy));
  }
  /* interp_a.Evaluate(_u,_v,&residue[0]) — utils.h:77 ; ceres Jet overload:
   * value from the scalar parts, derivative = dfdr * r.v + dfdc * c.v */
  double f, dfdr, dfdc;
  ea_oracle_bicubic(p->grid, p->grid_rows, p->grid_cols, u.a, v.a, &f, &dfdr, &dfdc);
  residue->a = f;
  for (int i = 0; i < NJ; ++i) residue->v[i] = dfdr * u.v[i] + dfdc * v.v[i];
  return 1;
}

int ea_oracle_block_jet(const ea_oracle_problem *p, const double q[4], const double t[3],
                        const double X[3], double *r, double jq[4], double jt[3]) {
  /* ceres: AutoDiffCostFunction<EAResidue,1,4,3> seeds Jet k with v[k] = 1 (utils.h:87-91) */
  jet qj[4], tj[3], res;
  for (int i = 0; i < 4; ++i) qj[i] = jet_var(q[i], i);
  for (int i = 0; i < 3; ++i) tj[i] = jet_var(t[i], 4 + i);
  if (!functor_jet(p, qj, tj, X, &res)) return 0;
  *r = res.a;
  for (int i = 0; i < 4; ++i) jq[i] = res.v[i];
  for (int i = 0; i < 3; ++i) jt[i] = res.v[4 + i];
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* ceres: QuaternionParameterization (local_parameterization.cc)                               */

void ea_oracle_quat_plus(const double x[4], const double delta[3], double x_plus_delta[4]) {
  const double norm_delta =
      sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (norm_delta > 0.0) {
    const double sin_delta_by_delta = sin(norm_delta) / norm_delta;
    double qd[4];
    qd[0] = cos(norm_delta);
    qd[1] = sin_delta_by_delta * delta[0];
    qd[2] = sin_delta_by_delta * delta[1];
    qd[3] = sin_delta_by_delta * delta[2];
    /* ceres: QuaternionProduct(q_delta, x, x_plus_delta) (rotation.h) */
    x_plus_delta[0] = qd[0] * x[0] - qd[1] * x[1] - qd[2] * x[2] - qd[3] * x[3];
    x_plus_delta[1] = qd[0] * x[1] + qd[1] * x[0] + qd[2] * x[3] - qd[3] * x[2];
    x_plus_delta[2] = qd[0] * x[2] - qd[1] * x[3] + qd[2] * x[0] + qd[3] * x[1];
    x_plus_delta[3] = qd[0] * x[3] + qd[1] * x[2] - qd[2] * x[1] + qd[3] * x[0];
  } else {
    for (int i = 0; i < 4; ++i) x_plus_delta[i] = x[i];
  }
}

void ea_oracle_quat_plus_jacobian(const double x[4], double P[12]) {
  P[0] = -x[1]; P[1] = -x[2]; P[2] = -x[3];
  P[3] = x[0];  P[4] = x[3];  P[5] = -x[2];
  P[6] = -x[3]; P[7] = x[0];  P[8] = x[1];
  P[9] = x[2];  P[10] = -x[1]; P[11] = x[0];
}

/* ------------------------------------------------------------------------------------------ */
/* analytic block: same residual, closed-form 1x6 row in the tangent space                     */

typedef struct {
  double R[3][3];
  double G[3][3][3]; /* G[j] = d R / d delta_j = sum_i dR/dq_i * P[i][j]  (general, |q| != 1 ok) */
} pose_consts;

static void pose_prepare(const ea_oracle_problem *p, const double q[4], pose_consts *pc) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  /* Eigen toRotationMatrix, same operation order as the Jet path above */
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double R[3][3];
  R[0][0] = 1.0 - (tyy + tzz); R[0][1] = txy - twz;         R[0][2] = txz + twy;
  R[1][0] = txy + twz;         R[1][1] = 1.0 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;         R[2][1] = tyz + twx;         R[2][2] = 1.0 - (txx + tyy);
  /* partials of R wrt (w,x,y,z) */
  const double dR[4][3][3] = {
      {{0, -2 * z, 2 * y}, {2 * z, 0, -2 * x}, {-2 * y, 2 * x, 0}},
      {{0, 2 * y, 2 * z}, {2 * y, -4 * x, -2 * w}, {2 * z, 2 * w, -4 * x}},
      {{-4 * y, 2 * x, 2 * w}, {2 * x, 0, 2 * z}, {-2 * w, 2 * z, -4 * y}},
      {{-4 * z, -2 * w, 2 * x}, {2 * w, -4 * z, 2 * y}, {2 * x, 2 * y, 0}}};
  double P[12];
  ea_oracle_quat_plus_jacobian(q, P);
  for (int j = 0; j < 3; ++j)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double s = 0.0;
        for (int i = 0; i < 4; ++i) s += dR[i][a][b] * P[3 * i + j];
        pc->G[j][a][b] = s;
      }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) pc->R[a][b] = R[a][b];
  if (p->rot_transposed) {
    for (int a = 0; a < 3; ++a)
      for (int b = a + 1; b < 3; ++b) {
        double tmp = pc->R[a][b];
        pc->R[a][b] = pc->R[b][a];
        pc->R[b][a] = tmp;
        for (int j = 0; j < 3; ++j) {
          tmp = pc->G[j][a][b];
          pc->G[j][a][b] = pc->G[j][b][a];
          pc->G[j][b][a] = tmp;
        }
      }
  }
}

static int block_analytic_pc(const ea_oracle_problem *p, const pose_consts *pc, const double t[3],
                             const double X[3], double *r, double j6[6]) {
  /* second camera: a' = T12inv [a;1], c = R a' + t, b = T12 [c;1]  (affine parts; the reference passes
   * rigid transforms whose last row is 0 0 0 1) */
  double a[3] = {X[0], X[1], X[2]};
  if (p->use_second_cam)
    for (int i = 0; i < 3; ++i)
      a[i] = p->T12inv[4 * i + 0] * X[0] + p->T12inv[4 * i + 1] * X[1] + p->T12inv[4 * i + 2] * X[2] + p->T12inv[4 * i + 3];
  double c[3], b[3];
  for (int i = 0; i < 3; ++i)
    c[i] = ((pc->R[i][0] * a[0] + pc->R[i][1] * a[1]) + pc->R[i][2] * a[2]) + t[i];
  if (p->use_second_cam)
    for (int i = 0; i < 3; ++i)
      b[i] = p->T12[4 * i + 0] * c[0] + p->T12[4 * i + 1] * c[1] + p->T12[4 * i + 2] * c[2] + p->T12[4 * i + 3];
  else
    for (int i = 0; i < 3; ++i) b[i] = c[i];
  if (p->z_guard > 0.0 && b[2] < p->z_guard && b[2] > -p->z_guard) return 0;
  const double bz = b[2] + p->z_eps;
  const double iz = 1.0 / bz;
  const double x = b[0] * iz, y = b[1] * iz;
  double u, v, xd_x = 1.0, xd_y = 0.0, yd_x = 0.0, yd_y = 1.0; /* d(distorted)/d(x,y) */
  if (p->use_distortion) {
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double D = 1.0 + p->k1 * r2 + p->k2 * r4 + p->k3 * r6;
    const double Dp = p->k1 + 2.0 * p->k2 * r2 + 3.0 * p->k3 * r4; /* dD/d(r2) */
    const double xd = x * D + 2.0 * p->p1 * x * y + p->p2 * (r2 + 2.0 * x * x);
    const double yd = y * D + 2.0 * p->p2 * x * y + p->p1 * (r2 + 2.0 * y * y);
    xd_x = D + 2.0 * x * x * Dp + 2.0 * p->p1 * y + 6.0 * p->p2 * x;
    xd_y = 2.0 * x * y * Dp + 2.0 * p->p1 * x + 2.0 * p->p2 * y;
    yd_x = 2.0 * x * y * Dp + 2.0 * p->p2 * y + 2.0 * p->p1 * x;
    yd_y = D + 2.0 * y * y * Dp + 2.0 * p->p2 * x + 6.0 * p->p1 * y;
    u = p->fx * xd + p->cx;
    v = p->fy * yd + p->cy;
  } else {
    u = p->fx * b[0] / bz + p->cx;
    v = p->fy * b[1] / bz + p->cy;
  }
  double f, Fu, Fv;
  ea_oracle_bicubic(p->grid, p->grid_rows, p->grid_cols, u, v, &f, &Fu, &Fv);
  *r = f;
  if (j6) {
    /* d r / d(x,y), then d r / d b, then (second camera) back through the affine map */
    const double rx = Fu * p->fx * xd_x + Fv * p->fy * yd_x;
    const double ry = Fu * p->fx * xd_y + Fv * p->fy * yd_y;
    double gb[3] = {rx * iz, ry * iz, -(rx * x + ry * y) * iz};
    double g[3];
    if (p->use_second_cam)
      for (int i = 0; i < 3; ++i) g[i] = p->T12[0 + i] * gb[0] + p->T12[4 + i] * gb[1] + p->T12[8 + i] * gb[2];
    else
      for (int i = 0; i < 3; ++i) g[i] = gb[i];
    for (int j = 0; j < 3; ++j) {
      double d0 = pc->G[j][0][0] * a[0] + pc->G[j][0][1] * a[1] + pc->G[j][0][2] * a[2];
      double d1 = pc->G[j][1][0] * a[0] + pc->G[j][1][1] * a[1] + pc->G[j][1][2] * a[2];
      double d2 = pc->G[j][2][0] * a[0] + pc->G[j][2][1] * a[1] + pc->G[j][2][2] * a[2];
      j6[j] = g[0] * d0 + g[1] * d1 + g[2] * d2;
    }
    j6[3] = g[0];
    j6[4] = g[1];
    j6[5] = g[2];
  }
  return 1;
}

int ea_oracle_block_analytic(const ea_oracle_problem *p, const double q[4], const double t[3],
                             const double X[3], double *r, double j6[6]) {
  pose_consts pc;
  pose_prepare(p, q, &pc);
  return block_analytic_pc(p, &pc, t, X, r, j6);
}

/* ------------------------------------------------------------------------------------------ */
/* ceres: loss_function.cc  (rho[0]=rho(s), rho[1]=rho'(s), rho[2]=rho''(s)),  s = r^2          */

static void loss_eval(int kind, double a, double s, double rho[3]) {
  if (kind == EA_ORACLE_LOSS_CAUCHY) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = b * log(sum);
    rho[1] = inv > DBL_MIN ? inv : DBL_MIN;
    rho[2] = -c * (inv * inv);
  } else if (kind == EA_ORACLE_LOSS_HUBER) {
    const double b = a * a;
    if (s > b) {
      const double r = sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = a / r > DBL_MIN ? a / r : DBL_MIN;
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* evaluator: what ceres::ProgramEvaluator produces for this problem shape                      */

int64_t ea_oracle_eval(const ea_oracle_problem *p, const double *xyz, int64_t n, int stride,
                       const double q[4], const double t[3], int jacobian_mode, double *cost,
                       double JtJ[36], double Jtr[6], double *r_out, double *J_out,
                       double *raw_r, double *raw_J) {
  pose_consts pc;
  pose_prepare(p, q, &pc);
  double P[12];
  ea_oracle_quat_plus_jacobian(q, P);
  double A[6][6], g[6], c = 0.0;
  memset(A, 0, sizeof(A));
  memset(g, 0, sizeof(g));
  int64_t n_invalid = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double *X = xyz + (size_t)i * (size_t)stride;
    double r, j6[6];
    int ok;
    if (jacobian_mode == EA_ORACLE_JAC_JET) {
      double jq[4], jt[3];
      ok = ea_oracle_block_jet(p, q, t, X, &r, jq, jt);
      if (ok) {
        /* ceres: local Jacobian = J_q (1x4) * P (4x3)   (residual_block.cc) */
        for (int k = 0; k < 3; ++k)
          j6[k] = jq[0] * P[k] + jq[1] * P[3 + k] + jq[2] * P[6 + k] + jq[3] * P[9 + k];
        j6[3] = jt[0]; j6[4] = jt[1]; j6[5] = jt[2];
      }
    } else {
      ok = block_analytic_pc(p, &pc, t, X, &r, j6);
    }
    if (!ok) {
      ++n_invalid;
      if (r_out) r_out[i] = NAN;
      if (raw_r) raw_r[i] = NAN;
      if (J_out) for (int k = 0; k < 6; ++k) J_out[6 * i + k] = NAN;
      if (raw_J) for (int k = 0; k < 6; ++k) raw_J[6 * i + k] = NAN;
      continue;
    }
    if (raw_r) raw_r[i] = r;
    if (raw_J) for (int k = 0; k < 6; ++k) raw_J[6 * i + k] = j6[k];
    /* ceres: cost += 0.5*rho[0]; Corrector: rho'' <= 0 for all three losses here =>
     * residual *= sqrt(rho'), jacobian *= sqrt(rho')  (corrector.cc) */
    double rho[3];
    loss_eval(p->loss_kind, p->loss_a, r * r, rho);
    c += 0.5 * rho[0];
    const double sq = sqrt(rho[1]);
    const double rc = sq * r;
    double jc[6];
    for (int k = 0; k < 6; ++k) jc[k] = sq * j6[k];
    if (r_out) r_out[i] = rc;
    if (J_out) for (int k = 0; k < 6; ++k) J_out[6 * i + k] = jc[k];
    for (int a = 0; a < 6; ++a) {
      g[a] += jc[a] * rc;
      for (int b = a; b < 6; ++b) A[a][b] += jc[a] * jc[b];
    }
  }
  if (cost) *cost = c;
  if (JtJ)
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) JtJ[6 * a + b] = a <= b ? A[a][b] : A[b][a];
  if (Jtr) for (int a = 0; a < 6; ++a) Jtr[a] = g[a];
  return n_invalid;
}

int64_t ea_oracle_cost(const ea_oracle_problem *p, const double *xyz, int64_t n, int stride,
                       const double q[4], const double t[3], double *cost) {
  pose_consts pc;
  pose_prepare(p, q, &pc);
  double c = 0.0;
  int64_t n_invalid = 0;
  for (int64_t i = 0; i < n; ++i) {
    double r;
    if (!block_analytic_pc(p, &pc, t, xyz + (size_t)i * (size_t)stride, &r, NULL)) {
      ++n_invalid;
      continue;
    }
    double rho[3];
    loss_eval(p->loss_kind, p->loss_a, r * r, rho);
    c += 0.5 * rho[0];
  }
  *cost = c;
  return n_invalid;
}

/* one ceres::Problem with several residual families sharing (q,t): sums over the terms;
 * r_out / J_out (nullable) are filled term after term */
static int64_t eval_terms_full(const ea_oracle_term *terms, int nterms, const double q[4], const double t[3],
                               int jacobian_mode, double *cost, double JtJ[36], double Jtr[6], double *r_out,
                               double *J_out) {
  double c = 0.0, A[36], g[6];
  memset(A, 0, sizeof(A));
  memset(g, 0, sizeof(g));
  int64_t bad = 0, off = 0;
  for (int k = 0; k < nterms; ++k) {
    double ck, Ak[36], gk[6];
    bad += ea_oracle_eval(terms[k].problem, terms[k].xyz, terms[k].n, terms[k].stride, q, t, jacobian_mode, &ck, Ak,
                          gk, r_out ? r_out + off : NULL, J_out ? J_out + 6 * off : NULL, NULL, NULL);
    c += ck;
    for (int i = 0; i < 36; ++i) A[i] += Ak[i];
    for (int i = 0; i < 6; ++i) g[i] += gk[i];
    off += terms[k].n;
  }
  if (cost) *cost = c;
  if (JtJ) memcpy(JtJ, A, sizeof(A));
  if (Jtr) memcpy(Jtr, g, sizeof(g));
  return bad;
}

int64_t ea_oracle_eval_terms(const ea_oracle_term *terms, int nterms, const double q[4], const double t[3],
                             int jacobian_mode, double *cost, double JtJ[36], double Jtr[6]) {
  return eval_terms_full(terms, nterms, q, t, jacobian_mode, cost, JtJ, Jtr, NULL, NULL);
}

static int64_t cost_terms(const ea_oracle_term *terms, int nterms, const double q[4], const double t[3], double *cost) {
  double c = 0.0;
  int64_t bad = 0;
  for (int k = 0; k < nterms; ++k) {
    double ck;
    bad += ea_oracle_cost(terms[k].problem, terms[k].xyz, terms[k].n, terms[k].stride, q, t, &ck);
    c += ck;
  }
  *cost = c;
  return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* linear algebra for the 6-dof step                                                            */

/* (A + diag(D^2)) y = g by Cholesky; returns 0 on failure.  A row-major 6x6 symmetric. */
static int solve_cholesky6(const double A[36], const double D[6], const double g[6], double y[6]) {
  double L[6][6];
  memset(L, 0, sizeof(L));
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j <= i; ++j) {
      double s = A[6 * i + j] + (i == j ? D[i] * D[i] : 0.0);
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return 0;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * z[k];
    z[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k][i] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 0; i < 6; ++i)
    if (!isfinite(y[i])) return 0;
  return 1;
}

/* ceres: DenseQRSolver — min || [J; diag(D)] y - [r; 0] ||  by Householder QR of the stacked
 * (n+6) x 6 matrix (column-major work array M, rhs b), J given row-major n x 6 with column
 * scaling S applied on the fly. */
static int solve_dense_qr(const double *J, const double *r, int64_t n, const double S[6],
                          const double D[6], double y[6]) {
  const int64_t m = n + 6;
  double *M = (double *)malloc(sizeof(double) * (size_t)m * 6);
  double *b = (double *)malloc(sizeof(double) * (size_t)m);
  if (!M || !b) { free(M); free(b); return 0; }
  for (int64_t i = 0; i < n; ++i) {
    for (int k = 0; k < 6; ++k) M[(size_t)k * m + i] = J[6 * i + k] * S[k];
    b[i] = r[i];
  }
  for (int k = 0; k < 6; ++k) {
    for (int j = 0; j < 6; ++j) M[(size_t)k * m + n + j] = (j == k) ? D[k] : 0.0;
    b[n + k] = 0.0;
  }
  int ok = 1;
  for (int k = 0; k < 6; ++k) {
    double *ck = M + (size_t)k * m;
    double nrm = 0.0;
    for (int64_t i = k; i < m; ++i) nrm += ck[i] * ck[i];
    nrm = sqrt(nrm);
    if (nrm == 0.0) { ok = 0; break; }
    const double alpha = ck[k] > 0 ? -nrm : nrm;
    const double vk = ck[k] - alpha; /* v = x - alpha e1 */
    double vnorm2 = vk * vk;
    for (int64_t i = k + 1; i < m; ++i) vnorm2 += ck[i] * ck[i];
    if (vnorm2 == 0.0) { ok = 0; break; }
    /* apply H = I - 2 v v^T / (v^T v) to the remaining columns and b */
    for (int j = k + 1; j < 6; ++j) {
      double *cj = M + (size_t)j * m;
      double dot = vk * cj[k];
      for (int64_t i = k + 1; i < m; ++i) dot += ck[i] * cj[i];
      const double f = 2.0 * dot / vnorm2;
      cj[k] -= f * vk;
      for (int64_t i = k + 1; i < m; ++i) cj[i] -= f * ck[i];
    }
    {
      double dot = vk * b[k];
      for (int64_t i = k + 1; i < m; ++i) dot += ck[i] * b[i];
      const double f = 2.0 * dot / vnorm2;
      b[k] -= f * vk;
      for (int64_t i = k + 1; i < m; ++i) b[i] -= f * ck[i];
    }
    ck[k] = alpha; /* R(k,k); below-diagonal entries of ck keep v (unused afterwards) */
  }
  if (ok) {
    for (int i = 5; i >= 0; --i) {
      double s = b[i];
      for (int k = i + 1; k < 6; ++k) s -= M[(size_t)k * m + i] * y[k];
      y[i] = s / M[(size_t)i * m + i];
      if (!isfinite(y[i])) ok = 0;
    }
  }
  free(M);
  free(b);
  return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* ceres: TrustRegionMinimizer::Minimize with LevenbergMarquardtStrategy (or traditional
 * DoglegStrategy), monotonic steps, Jacobi scaling, TrustRegionStepEvaluator(…, 0).
 * ref call site: standalone_edge_align.cpp:282-286 ; ROS flavour src/SolveEA.cpp:184-198     */

typedef struct {
  double radius, decrease_factor;
  int reuse_diagonal;
  double diagonal[6];
  /* dogleg state */
  double mu, alpha, dogleg_step_norm;
  double dl_diag[6], dl_grad[6], dl_gn[6];
  int dl_reuse;
} tr_strategy;

static double vec_norm(const double *v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}

static void plus7(const double x[7], const double delta[6], double out[7]) {
  ea_oracle_quat_plus(x, delta, out);
  for (int i = 0; i < 3; ++i) out[4 + i] = x[4 + i] + delta[3 + i];
}

typedef struct {
  /* scaled-space quantities at the current x (Js = J S) */
  double A[36];  /* Js^T Js */
  double g[6];   /* Js^T r   (scaled gradient) */
  double *Jc, *rc; /* materialised corrected J (unscaled, n x 6) and r, only for DENSE_QR */
} lin_state;

static int lm_compute_step(tr_strategy *st, const ea_oracle_options *opt, const lin_state *ls,
                           int64_t n, const double S[6], double step[6]) {
  /* ceres: LevenbergMarquardtStrategy::ComputeStep */
  if (!st->reuse_diagonal) {
    for (int i = 0; i < 6; ++i) {
      double d = ls->A[6 * i + i];
      d = d < opt->min_lm_diagonal ? opt->min_lm_diagonal : d;
      d = d > opt->max_lm_diagonal ? opt->max_lm_diagonal : d;
      st->diagonal[i] = d;
    }
  }
  double D[6];
  for (int i = 0; i < 6; ++i) D[i] = sqrt(st->diagonal[i] / st->radius);
  double y[6];
  int ok;
  if (opt->linear_solver == EA_ORACLE_LIN_DENSE_QR && ls->Jc)
    ok = solve_dense_qr(ls->Jc, ls->rc, n, S, D, y);
  else
    ok = solve_cholesky6(ls->A, D, ls->g, y);
  st->reuse_diagonal = 1;
  if (!ok) return 0;
  for (int i = 0; i < 6; ++i) step[i] = -y[i]; /* solve J y = r, x = -y */
  return 1;
}

static int dogleg_compute_step(tr_strategy *st, const ea_oracle_options *opt, const lin_state *ls,
                               double step[6]) {
  /* ceres: DoglegStrategy::ComputeStep, TRADITIONAL_DOGLEG (dogleg_strategy.cc) */
  if (!st->dl_reuse) {
    for (int i = 0; i < 6; ++i) {
      double d = ls->A[6 * i + i];
      d = d < opt->min_lm_diagonal ? opt->min_lm_diagonal : d;
      d = d > opt->max_lm_diagonal ? opt->max_lm_diagonal : d;
      st->dl_diag[i] = sqrt(d);
    }
    for (int i = 0; i < 6; ++i) st->dl_grad[i] = ls->g[i] / st->dl_diag[i];
    /* Cauchy point: alpha = |g|^2 / |J (g ./ diag)|^2 */
    double gs[6], q = 0.0;
    for (int i = 0; i < 6; ++i) gs[i] = st->dl_grad[i] / st->dl_diag[i];
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) q += gs[a] * ls->A[6 * a + b] * gs[b];
    double g2 = 0.0;
    for (int i = 0; i < 6; ++i) g2 += st->dl_grad[i] * st->dl_grad[i];
    st->alpha = g2 / q;
    /* Gauss-Newton step with mu regularisation */
    int ok = 0;
    const double mu_increase = 10.0, max_mu = 1.0, min_mu = 1e-8;
    while (st->mu < max_mu) {
      double D[6], y[6];
      for (int i = 0; i < 6; ++i) D[i] = st->dl_diag[i] * sqrt(st->mu);
      ok = solve_cholesky6(ls->A, D, ls->g, y);
      if (ok) {
        for (int i = 0; i < 6; ++i) st->dl_gn[i] = y[i];
        break;
      }
      st->mu *= mu_increase;
    }
    if (!ok) return 0;
    st->mu = fmax(min_mu, 2.0 * st->mu / mu_increase);
    for (int i = 0; i < 6; ++i) st->dl_gn[i] *= -st->dl_diag[i];
  }
  const double gn_norm = vec_norm(st->dl_gn, 6);
  if (gn_norm <= st->radius) {
    for (int i = 0; i < 6; ++i) step[i] = st->dl_gn[i] / st->dl_diag[i];
    st->dogleg_step_norm = gn_norm;
    return 1;
  }
  const double gradient_norm = vec_norm(st->dl_grad, 6);
  if (gradient_norm * st->alpha >= st->radius) {
    for (int i = 0; i < 6; ++i)
      step[i] = -(st->radius / gradient_norm) * st->dl_grad[i] / st->dl_diag[i];
    st->dogleg_step_norm = st->radius;
    return 1;
  }
  double b_dot_a = 0.0;
  for (int i = 0; i < 6; ++i) b_dot_a += -st->alpha * st->dl_grad[i] * st->dl_gn[i];
  const double a_sq = (st->alpha * gradient_norm) * (st->alpha * gradient_norm);
  const double bma_sq = a_sq - 2.0 * b_dot_a + gn_norm * gn_norm;
  const double c = b_dot_a - a_sq;
  const double d = sqrt(c * c + bma_sq * (st->radius * st->radius - a_sq));
  const double beta = (c <= 0) ? (d - c) / bma_sq : (st->radius * st->radius - a_sq) / (d + c);
  double dl[6];
  for (int i = 0; i < 6; ++i)
    dl[i] = (-st->alpha * (1.0 - beta)) * st->dl_grad[i] + beta * st->dl_gn[i];
  st->dogleg_step_norm = vec_norm(dl, 6);
  for (int i = 0; i < 6; ++i) step[i] = dl[i] / st->dl_diag[i];
  return 1;
}

int ea_oracle_solve(const ea_oracle_problem *p, const double *xyz, int64_t n, int stride,
                    const ea_oracle_options *opt, double q[4], double t[3],
                    ea_oracle_summary *sum) {
  ea_oracle_term term = {p, xyz, n, stride};
  return ea_oracle_solve_terms(&term, 1, opt, q, t, sum);
}

int ea_oracle_solve_terms(const ea_oracle_term *terms, int nterms, const ea_oracle_options *opt, double q[4],
                          double t[3], ea_oracle_summary *sum) {
  memset(sum, 0, sizeof(*sum));
  int64_t n = 0;
  for (int k = 0; k < nterms; ++k) n += terms[k].n;
  const int need_J = (opt->linear_solver == EA_ORACLE_LIN_DENSE_QR);
  double *Jc = NULL, *rc = NULL;
  if (need_J) {
    Jc = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * 6);
    rc = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  }
  double x[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
  double x_cost, JtJ[36], Jtr[6];
  double S[6] = {1, 1, 1, 1, 1, 1};
  lin_state ls;
  ls.Jc = Jc;
  ls.rc = rc;
  tr_strategy st;
  memset(&st, 0, sizeof(st));
  st.radius = opt->initial_trust_region_radius;
  st.decrease_factor = 2.0;
  st.reuse_diagonal = 0;
  st.mu = 1e-8;
  int it = 0;
  int rcode = EA_ORACLE_NO_CONVERGENCE, why = EA_ORACLE_WHY_NONE;

/* gradient_max_norm = || x - Plus(x, -gradient) ||_inf in the ambient space
 * (ceres: TrustRegionMinimizer::EvaluateGradientAndJacobian) */
#define EVAL_AT_X()                                                                             \
  do {                                                                                          \
    int64_t bad = eval_terms_full(terms, nterms, x, x + 4, opt->jacobian_mode, &x_cost, JtJ,    \
                                  Jtr, rc, Jc);                                                 \
    sum->num_residual_evals += n;                                                               \
    sum->num_jacobian_evals += n;                                                               \
    eval_ok = (bad == 0) && (fabs(x_cost) <= DBL_MAX); /* IsArrayValid: non-finite residuals fail */ \
  } while (0)

  int eval_ok;
  EVAL_AT_X();
  if (!eval_ok) {
    sum->termination = EA_ORACLE_FAILURE;
    sum->why = EA_ORACLE_WHY_INITIAL_EVAL_FAILED;
    free(Jc); free(rc);
    return sum->termination;
  }
  if (opt->jacobi_scaling) /* computed once, at iteration 0 */
    for (int i = 0; i < 6; ++i) S[i] = 1.0 / (1.0 + sqrt(JtJ[6 * i + i]));

#define SCALE_SYSTEM()                                                       \
  do {                                                                       \
    for (int a = 0; a < 6; ++a) {                                            \
      ls.g[a] = Jtr[a] * S[a];                                               \
      for (int b = 0; b < 6; ++b) ls.A[6 * a + b] = JtJ[6 * a + b] * S[a] * S[b]; \
    }                                                                        \
  } while (0)
#define GRAD_NORMS(out_max)                                                  \
  do {                                                                       \
    double neg[6], xp[7], m = 0.0;                                           \
    for (int i = 0; i < 6; ++i) neg[i] = -Jtr[i];                            \
    plus7(x, neg, xp);                                                       \
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(x[i] - xp[i]));             \
    out_max = m;                                                             \
  } while (0)

  SCALE_SYSTEM();
  double x_norm = vec_norm(x, 7);
  double gmax;
  GRAD_NORMS(gmax);
  sum->initial_cost = x_cost;
  sum->it_cost[0] = x_cost;
  sum->it_gradient_max_norm[0] = gmax;
  sum->it_radius[0] = st.radius;
  sum->it_successful[0] = 1;
  if (opt->verbose)
    printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n"
           "%4d  %.6e    0.00e+00    %.2e   0.00e+00   0.00e+00  %.2e\n", 0, x_cost, gmax, st.radius);
  int num_consecutive_invalid = 0;

  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (it >= opt->max_num_iterations) { rcode = EA_ORACLE_NO_CONVERGENCE; why = EA_ORACLE_WHY_MAX_ITERATIONS; break; }
    if (sum->it_gradient_max_norm[it] <= opt->gradient_tolerance) { rcode = EA_ORACLE_CONVERGENCE; why = EA_ORACLE_WHY_GRADIENT_TOL; break; }
    if (st.radius <= opt->min_trust_region_radius) { rcode = EA_ORACLE_CONVERGENCE; why = EA_ORACLE_WHY_MIN_RADIUS; break; }
    if (it + 1 >= EA_ORACLE_MAX_ITERS) { rcode = EA_ORACLE_NO_CONVERGENCE; why = EA_ORACLE_WHY_MAX_ITERATIONS; break; }
    ++it;
    sum->it_gradient_max_norm[it] = sum->it_gradient_max_norm[it - 1];
    sum->it_cost[it] = x_cost;

    /* ComputeTrustRegionStep */
    double step_s[6], delta[6];
    int step_ok = (opt->strategy == EA_ORACLE_STRATEGY_DOGLEG)
                      ? dogleg_compute_step(&st, opt, &ls, step_s)
                      : lm_compute_step(&st, opt, &ls, n, S, step_s);
    double model_cost_change = 0.0;
    if (step_ok) {
      /* model_cost_change = -(Js s)^T (r + Js s / 2) = -(g^T s + s^T A s / 2) */
      double gs = 0.0, sAs = 0.0;
      for (int a = 0; a < 6; ++a) {
        gs += ls.g[a] * step_s[a];
        for (int b = 0; b < 6; ++b) sAs += step_s[a] * ls.A[6 * a + b] * step_s[b];
      }
      model_cost_change = -(gs + 0.5 * sAs);
      if (!(model_cost_change > 0.0)) step_ok = 0;
    }
    if (!step_ok) {
      /* HandleInvalidStep */
      sum->it_successful[it] = 0;
      sum->it_radius[it] = st.radius;
      ++sum->num_unsuccessful_steps;
      if (++num_consecutive_invalid >= opt->max_num_consecutive_invalid_steps) {
        rcode = EA_ORACLE_FAILURE; why = EA_ORACLE_WHY_TOO_MANY_INVALID_STEPS; break;
      }
      if (opt->strategy == EA_ORACLE_STRATEGY_DOGLEG) { st.mu *= 10.0; st.dl_reuse = 0; }
      else { st.radius *= 0.5; st.reuse_diagonal = 1; }
      continue;
    }
    num_consecutive_invalid = 0;
    for (int i = 0; i < 6; ++i) delta[i] = step_s[i] * S[i];

    /* ComputeCandidatePointAndEvaluateCost */
    double cand[7], cand_cost;
    plus7(x, delta, cand);
    {
      int64_t bad = cost_terms(terms, nterms, cand, cand + 4, &cand_cost);
      sum->num_residual_evals += n;
      /* "Step failed to evaluate. Treating it as a step with infinite cost" (functor false, or non-finite residuals) */
      if (bad || !(fabs(cand_cost) <= DBL_MAX)) cand_cost = DBL_MAX;
    }
    /* ParameterToleranceReached */
    double dx[7];
    for (int i = 0; i < 7; ++i) dx[i] = x[i] - cand[i];
    const double step_norm = vec_norm(dx, 7);
    sum->it_step_norm[it] = step_norm;
    if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      sum->it_radius[it] = st.radius;
      rcode = EA_ORACLE_CONVERGENCE; why = EA_ORACLE_WHY_PARAMETER_TOL; break;
    }
    /* FunctionToleranceReached */
    const double cost_change = x_cost - cand_cost;
    sum->it_cost_change[it] = cost_change;
    if (fabs(cost_change) <= opt->function_tolerance * x_cost) {
      sum->it_radius[it] = st.radius;
      rcode = EA_ORACLE_CONVERGENCE; why = EA_ORACLE_WHY_FUNCTION_TOL; break;
    }
    /* IsStepSuccessful: monotonic TrustRegionStepEvaluator => quality = cost_change / model_cost_change */
    const double relative_decrease = cost_change / model_cost_change;
    sum->it_relative_decrease[it] = relative_decrease;
    if (relative_decrease > opt->min_relative_decrease) {
      /* HandleSuccessfulStep */
      memcpy(x, cand, sizeof(x));
      x_norm = vec_norm(x, 7);
      EVAL_AT_X();
      if (!eval_ok) { rcode = EA_ORACLE_FAILURE; why = EA_ORACLE_WHY_EVAL_FAILED; break; }
      SCALE_SYSTEM();
      GRAD_NORMS(gmax);
      sum->it_gradient_max_norm[it] = gmax;
      sum->it_cost[it] = x_cost;
      sum->it_successful[it] = 1;
      ++sum->num_successful_steps;
      if (opt->strategy == EA_ORACLE_STRATEGY_DOGLEG) {
        if (relative_decrease < 0.25) st.radius *= 0.5;
        if (relative_decrease > 0.75) st.radius = fmax(st.radius, 3.0 * st.dogleg_step_norm);
        st.radius = fmin(st.radius, opt->max_trust_region_radius);
        st.dl_reuse = 0;
      } else {
        const double f = 2.0 * relative_decrease - 1.0;
        st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - f * f * f);
        st.radius = fmin(opt->max_trust_region_radius, st.radius);
        st.decrease_factor = 2.0;
        st.reuse_diagonal = 0;
      }
    } else {
      /* HandleUnsuccessfulStep */
      sum->it_successful[it] = 0;
      ++sum->num_unsuccessful_steps;
      if (opt->strategy == EA_ORACLE_STRATEGY_DOGLEG) {
        st.radius *= 0.5;
        st.dl_reuse = 1;
      } else {
        st.radius = st.radius / st.decrease_factor;
        st.decrease_factor *= 2.0;
        st.reuse_diagonal = 1;
      }
    }
    sum->it_radius[it] = st.radius;
    if (opt->verbose)
      printf("%4d  %.6e  % .2e    %.2e   %.2e  % .2e  %.2e\n", it, x_cost, cost_change,
             sum->it_gradient_max_norm[it], step_norm, relative_decrease, st.radius);
  }
#undef EVAL_AT_X
#undef SCALE_SYSTEM
#undef GRAD_NORMS
  for (int i = 0; i < 4; ++i) q[i] = x[i];
  for (int i = 0; i < 3; ++i) t[i] = x[4 + i];
  sum->termination = rcode;
  sum->why = why;
  sum->num_iterations = it;
  sum->final_cost = x_cost;
  free(Jc);
  free(rc);
  return rcode;
}

// ea_lm.h — the 6-dof trust-region state machine, compiled for both the host and gfx950.
//
// It replaces what ceres::Solve does around the evaluator for this problem shape
// (ref: standalone/standalone_edge_align.cpp:277-286, src/SolveEA.cpp:180-198):
// QuaternionParameterization::Plus, Jacobi scaling, LevenbergMarquardtStrategy (or traditional
// DoglegStrategy), the monotonic step evaluator, and TrustRegionMinimizer's convergence tests.
// The N x 6 Jacobian is never materialised: the state machine only consumes the 32 fp64
// accumulators (JtJ, Jtr, cost, #invalid) the evaluation kernels reduce.
//
// One evaluation per iteration: every evaluation computes cost AND JtJ/Jtr at the candidate
// pose, so an accepted step needs no second pass (Ceres evaluates residuals at the candidate and
// then residuals+Jacobian again at the same point; the numbers are identical).
#pragma once
#include <float.h>
#include <math.h>

#include "ea_types.h"

#ifndef EA_LM_PROBE
#define EA_LM_PROBE(k) do {} while (0)  // diagnostic build only (scripts/build_stamps.sh)
#endif

// Branch-layout hints for the serial state machine: it runs cold on one lane (a different CU every launch), so the
// usual path -- step accepted, nothing converged, linear solve fine -- should be the fall-through one that the
// sequential instruction prefetch covers.
#define EA_LIKELY(x) __builtin_expect(!!(x), 1)
#define EA_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace ea {

struct LMOptions {
  int max_num_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  int max_num_consecutive_invalid_steps;
  int jacobi_scaling;
  int strategy;  // 0 LM, 1 traditional dogleg
};

constexpr int kTrace = 128;

// Hot state: what every iteration touches.  The LM-step kernel keeps it in registers on lane 0.
struct LMState {
  double x[7], cand[7];
  double x_norm, cost;
  double S[6];         // Jacobi column scaling, fixed at iteration 0
  // LevenbergMarquardtStrategy
  double radius, decrease_factor, diagonal[6];
  // DoglegStrategy (traditional) scalars
  double mu, alpha, dogleg_step_norm;
  double model_cost_change;
  double gradient_max_norm;
  int reuse_diagonal, dl_reuse;
  int iteration;
  int running;      // 1 while the solve needs another evaluation
  int termination;  // ea_termination
  int why;          // ea_why
  int num_successful, num_unsuccessful, num_consecutive_invalid;
  int num_evals;    // whole-problem evaluations performed
  int rot_transposed;
  int pad_;
};

// Cold state: written once per accepted step, read back only after a rejected / invalid step (the system at x is
// then reused with a smaller radius) and by the dogleg strategy.  Lives in device memory, never staged.
struct LMCold {
  double A[21], g[6];  // unscaled JtJ (upper triangle, row-major packed like the accumulator slots) and Jtr at x
  double dl_diag[6], dl_grad[6], dl_gn[6];
};

// member-by-member copy (the step kernel's register copy of the state: a whole-struct copy left the members the LM strategy
// never touches -- mu, alpha, dogleg_step_norm -- in a private array, which the compiler placed in LDS behind a read of the
// dispatch packet for the workgroup's shape: a scalar load from the queue's ring buffer on the state machine's critical path)
static_assert(sizeof(LMState) == 35 * sizeof(double) + 12 * sizeof(int), "lm_copy_state copies member by member: a new member goes there too");
EA_HD inline void lm_copy_state(LMState *d, const LMState *s) {
  for (int i = 0; i < 7; ++i) { d->x[i] = s->x[i]; d->cand[i] = s->cand[i]; }
  d->x_norm = s->x_norm; d->cost = s->cost;
  for (int i = 0; i < 6; ++i) { d->S[i] = s->S[i]; d->diagonal[i] = s->diagonal[i]; }
  d->radius = s->radius; d->decrease_factor = s->decrease_factor;
  d->mu = s->mu; d->alpha = s->alpha; d->dogleg_step_norm = s->dogleg_step_norm;
  d->model_cost_change = s->model_cost_change; d->gradient_max_norm = s->gradient_max_norm;
  d->reuse_diagonal = s->reuse_diagonal; d->dl_reuse = s->dl_reuse; d->iteration = s->iteration; d->running = s->running;
  d->termination = s->termination; d->why = s->why; d->num_successful = s->num_successful; d->num_unsuccessful = s->num_unsuccessful;
  d->num_consecutive_invalid = s->num_consecutive_invalid; d->num_evals = s->num_evals; d->rot_transposed = s->rot_transposed;
  d->pad_ = s->pad_;
}

// packed upper-triangle index of (a,b), a <= b, 6x6
EA_HD constexpr int sym6(int a, int b) { return a <= b ? a * 6 - a * (a - 1) / 2 + (b - a) : b * 6 - b * (b - 1) / 2 + (a - b); }

// per-iteration trace (what minimizer_progress_to_stdout would print); written once per
// iteration, never read back by the state machine
struct LMTrace {
  double it_cost[kTrace], it_cost_change[kTrace], it_gradient_max_norm[kTrace];
  double it_step_norm[kTrace], it_relative_decrease[kTrace], it_radius[kTrace];
  int it_successful[kTrace];
};

// ---- small helpers ---------------------------------------------------------------------------

EA_HD inline double norm_n(const double *v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}

// QuaternionParameterization::Plus: x_plus = [cos|d|, sin|d|/|d| d] (x) x
// `small_expected`: which of the two forms below is laid out as the fall-through path -- a trust-region step is small, the
// (minus) gradient that Ceres pushes through Plus for its gradient_max_norm is not
EA_HD inline void quat_plus(const double x[4], const double d[3], double out[4], bool small_expected = true) {
  const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  if (EA_LIKELY(n2 > 0.0)) {
    double s, cs;  // s = sin|d| / |d|, cs = cos|d|
    const bool small = n2 < 0.25;
    if (small_expected ? EA_LIKELY(small) : EA_UNLIKELY(small)) {
      // |d| < 0.5 rad -- every step of a converging solve: both are even functions of |d|, evaluated as Taylor polynomials in
      // |d|^2 (relative error < 1e-16 on this range, checked against 80-bit sin / cos) -- no square root, no argument
      // reduction, no division: ~20 dependent-free FMAs instead of ~120 instructions, twice per iteration on the one
      // lane every LM iteration waits for.
      s = n2 * 2.8114572543455206e-15 - 7.6471637318198164e-13;
      s = s * n2 + 1.6059043836821613e-10;
      s = s * n2 - 2.505210838544172e-08;
      s = s * n2 + 2.7557319223985893e-06;
      s = s * n2 - 0.00019841269841269841;
      s = s * n2 + 0.0083333333333333332;
      s = s * n2 - 0.16666666666666666;
      s = s * n2 + 1.0;
      cs = n2 * -1.5619206968586225e-16 + 4.7794773323873853e-14;
      cs = cs * n2 - 1.1470745597729725e-11;
      cs = cs * n2 + 2.08767569878681e-09;
      cs = cs * n2 - 2.7557319223985888e-07;
      cs = cs * n2 + 2.4801587301587302e-05;
      cs = cs * n2 - 0.0013888888888888889;
      cs = cs * n2 + 0.041666666666666664;
      cs = cs * n2 - 0.5;
      cs = cs * n2 + 1.0;
    } else {
      const double nd = sqrt(n2);
      double sn;
      sincos(nd, &sn, &cs);
      s = sn / nd;
    }
    const double q0 = cs, q1 = s * d[0], q2 = s * d[1], q3 = s * d[2];
    out[0] = q0 * x[0] - q1 * x[1] - q2 * x[2] - q3 * x[3];
    out[1] = q0 * x[1] + q1 * x[0] + q2 * x[3] - q3 * x[2];
    out[2] = q0 * x[2] - q1 * x[3] + q2 * x[0] + q3 * x[1];
    out[3] = q0 * x[3] + q1 * x[2] - q2 * x[1] + q3 * x[0];
  } else {
    for (int i = 0; i < 4; ++i) out[i] = x[i];
  }
}

EA_HD inline void pose_plus(const double x[7], const double delta[6], double out[7], bool small_expected = true) {
  quat_plus(x, delta, out, small_expected);
  for (int i = 0; i < 3; ++i) out[4 + i] = x[4 + i] + delta[3 + i];
}

// Pose constants for the kernels.  R follows Eigen's un-normalised toRotationMatrix
// (ref: standalone/utils.h:51-53); G_j = sum_i dR/dq_i P_ij is the exact tangent derivative for
// any |q| (P = QuaternionParameterization Jacobian); for |q| = 1 it reduces to -2 [R a]x and the
// kernels take the short form.
EA_HD inline void make_pose_core(const double x[7], int rot_transposed, int active,
                                 PoseState *ps, bool zero_unused_G = true) {
  const double w = x[0], qx = x[1], qy = x[2], qz = x[3];
  for (int i = 0; i < 4; ++i) ps->q[i] = x[i];
  for (int i = 0; i < 3; ++i) ps->t[i] = x[4 + i];
  double R[9];
  R[0] = 1.0 - 2.0 * (qy * qy + qz * qz); R[1] = 2.0 * (qx * qy - w * qz); R[2] = 2.0 * (qx * qz + w * qy);
  R[3] = 2.0 * (qx * qy + w * qz); R[4] = 1.0 - 2.0 * (qx * qx + qz * qz); R[5] = 2.0 * (qy * qz - w * qx);
  R[6] = 2.0 * (qx * qz - w * qy); R[7] = 2.0 * (qy * qz + w * qx); R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
  const double n2 = w * w + qx * qx + qy * qy + qz * qz;
  const int unit_q = (fabs(n2 - 1.0) <= 1e-12 && !rot_transposed) ? 1 : 0;
  if (EA_UNLIKELY(!unit_q)) {
    const double dR[4][9] = {
        {0, -2 * qz, 2 * qy, 2 * qz, 0, -2 * qx, -2 * qy, 2 * qx, 0},
        {0, 2 * qy, 2 * qz, 2 * qy, -4 * qx, -2 * w, 2 * qz, 2 * w, -4 * qx},
        {-4 * qy, 2 * qx, 2 * w, 2 * qx, 0, 2 * qz, -2 * w, 2 * qz, -4 * qy},
        {-4 * qz, -2 * w, 2 * qx, 2 * w, -4 * qz, 2 * qy, 2 * qx, 2 * qy, 0}};
    const double P[12] = {-qx, -qy, -qz, w, qz, -qy, -qz, w, qx, qy, -qx, w};
    for (int j = 0; j < 3; ++j)
      for (int e = 0; e < 9; ++e) {
        double s = 0.0;
        for (int i = 0; i < 4; ++i) s += dR[i][e] * P[3 * i + j];
        ps->G[9 * j + e] = s;
      }
  } else if (zero_unused_G) {
    for (int e = 0; e < 27; ++e) ps->G[e] = 0.0;  // never read: the kernels take the -2 [R a]x form
  }
  for (int e = 0; e < 9; ++e) ps->R[e] = R[e];
  if (rot_transposed) {  // ref: include/EAResidue.h:99-101 applies R^T
    for (int a = 0; a < 3; ++a)
      for (int b = a + 1; b < 3; ++b) {
        double tmp = ps->R[3 * a + b];
        ps->R[3 * a + b] = ps->R[3 * b + a];
        ps->R[3 * b + a] = tmp;
        for (int j = 0; j < 3; ++j) {
          tmp = ps->G[9 * j + 3 * a + b];
          ps->G[9 * j + 3 * a + b] = ps->G[9 * j + 3 * b + a];
          ps->G[9 * j + 3 * b + a] = tmp;
        }
      }
  }
  ps->unit_q = unit_q;
  ps->active = active;
}

// float mirrors for the fp32 kernels (the LM-step kernel computes them lane-parallel instead)
EA_HD inline void make_pose_state(const double x[7], int rot_transposed, int active,
                                  PoseState *ps) {
  make_pose_core(x, rot_transposed, active, ps);
  for (int e = 0; e < 9; ++e) ps->Rf[e] = (float)ps->R[e];
  for (int e = 0; e < 27; ++e) ps->Gf[e] = (float)ps->G[e];
  for (int e = 0; e < 3; ++e) ps->tf[e] = (float)ps->t[e];
}

// Reciprocal for pivots and radii.  On the device: v_rcp_f64 + two Newton steps (5 dependent instructions instead of
// the ~14 of an IEEE division; within 1 ulp) -- this code runs on one lane and is pure latency.
EA_HD inline double ea_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
#else
  return 1.0 / x;
#endif
}

// (A + diag(D2)) y = g for the 6x6 normal equations, square-root-free Cholesky (A + D2 = L diag(d) L^T, unit L).
// Returns false when not positive definite / not finite.  Six dependent reciprocals are the whole serial chain.
EA_HD inline bool solve_spd6(const double A[21] /* packed upper */, const double D2[6], const double g[6], double y[6]) {
  double L[36], M[36], inv[6];  // M_ik = L_ik d_k
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[sym6(j, j)] + D2[j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * M[6 * j + k];
    if (EA_UNLIKELY(!(d > 0.0))) return false;
    inv[j] = ea_rcp(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[sym6(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= M[6 * i + k] * L[6 * j + k];
      M[6 * i + j] = t;
      L[6 * i + j] = t * inv[j];
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[6 * i + k] * z[k];
    z[i] = s;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = z[i] * inv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (EA_UNLIKELY(!(fabs(y[i]) <= DBL_MAX))) return false;
  return true;
}

EA_HD inline void lm_trace(const LMState *s, LMTrace *tr, int it, double cost_change, double step_norm,
                           double rel, int successful) {
  if (tr && it < kTrace) {
    tr->it_cost[it] = s->cost;
    tr->it_cost_change[it] = cost_change;
    tr->it_gradient_max_norm[it] = s->gradient_max_norm;
    tr->it_step_norm[it] = step_norm;
    tr->it_relative_decrease[it] = rel;
    tr->it_radius[it] = s->radius;
    tr->it_successful[it] = successful;
  }
}

// Global-memory writes of one state-machine call, deferred to its end: on the device every store issued before the
// arithmetic would sit in the same in-order memory counter the compiler later waits on, and put the store round trip
// back on the critical path of a single lane.
struct LMPending {
  int store_system;  // 1: the cold state takes JtJ / Jtr from the accumulator slots (evaluation accepted at x)
  int trace_it;      // >= 0: trace row to write
  int successful;
  double cost, cost_change, gradient_max_norm, step_norm, rel, radius;
};

EA_HD inline void lm_pend_trace(const LMState *s, LMPending *p, int it, double cost_change, double step_norm,
                                double rel, int successful) {
  p->trace_it = it;
  p->successful = successful;
  p->cost = s->cost;
  p->cost_change = cost_change;
  p->gradient_max_norm = s->gradient_max_norm;
  p->step_norm = step_norm;
  p->rel = rel;
  p->radius = s->radius;
}

EA_HD inline void lm_flush(const LMPending *p, LMCold *c, LMTrace *tr, const double acc[kAccSlots]) {
  if (EA_LIKELY(p->store_system)) {
#pragma unroll
    for (int k = 0; k < 21; ++k) c->A[k] = acc[kAccJtJ + k];
#pragma unroll
    for (int a = 0; a < 6; ++a) c->g[a] = acc[kAccJtr + a];
  }
  if (EA_LIKELY(tr && p->trace_it >= 0 && p->trace_it < kTrace)) {
    const int it = p->trace_it;
    tr->it_cost[it] = p->cost;
    tr->it_cost_change[it] = p->cost_change;
    tr->it_gradient_max_norm[it] = p->gradient_max_norm;
    tr->it_step_norm[it] = p->step_norm;
    tr->it_relative_decrease[it] = p->rel;
    tr->it_radius[it] = p->radius;
    tr->it_successful[it] = p->successful;
  }
}

EA_HD inline void lm_finish(LMState *s, int termination, int why) {
  s->running = 0;
  s->termination = termination;
  s->why = why;
}

EA_HD inline void lm_init(LMState *s, const LMOptions *o, const double q[4], const double t[3],
                          int rot_transposed) {
  for (int i = 0; i < 4; ++i) s->x[i] = q[i];
  for (int i = 0; i < 3; ++i) s->x[4 + i] = t[i];
  for (int i = 0; i < 7; ++i) s->cand[i] = s->x[i];
  s->x_norm = norm_n(s->x, 7);
  s->cost = 0.0;
  for (int i = 0; i < 6; ++i) { s->S[i] = 1.0; s->diagonal[i] = 0.0; }
  s->radius = o->initial_trust_region_radius;
  s->decrease_factor = 2.0;
  s->reuse_diagonal = 0;
  s->mu = 1e-8; s->alpha = 0.0; s->dogleg_step_norm = 0.0; s->dl_reuse = 0;
  s->model_cost_change = 0.0;
  s->gradient_max_norm = 0.0;
  s->iteration = 0;
  s->running = 1;
  s->termination = 1; s->why = 0;
  s->num_successful = s->num_unsuccessful = s->num_consecutive_invalid = 0;
  s->num_evals = 0;
  s->rot_transposed = rot_transposed;
  s->pad_ = 0;
}

// The evaluation at s->x delivered `acc`: take the cost and
// gradient_max_norm = || x - Plus(x, -gradient) ||_inf  (ambient space); the unscaled system itself goes to the
// cold state in lm_flush.
// LITE (device only, ea_lm_iter_kernel): the caller wants the next candidate pose and nothing else -- whatever only feeds the
// stored state, the trace or the gradient-tolerance test is left out (gradient_max_norm's Plus() through sqrt / sincos /
// division, the norm of the accepted x).  Every workgroup of an evaluation runs this form; one more runs the full one and
// owns the state, so a solve that ends on the gradient tolerance merely finds one evaluation launched in vain.
template <bool LITE = false>
EA_HD inline void lm_take_system(LMState *s, LMPending *pend, const double acc[kAccSlots]) {
  pend->store_system = 1;
  s->cost = acc[kAccCost];
  if constexpr (LITE) return;
  double neg[6], xp[7], m = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) neg[i] = -acc[kAccJtr + i];
  pose_plus(s->x, neg, xp, /*small_expected=*/false);
#pragma unroll
  for (int i = 0; i < 7; ++i) m = fmax(m, fabs(s->x[i] - xp[i]));
  s->gradient_max_norm = m;
}

// scaled-space step from the current strategy; false = linear solve failed.  As: packed upper triangle.
// STRAT: trust-region strategy as a compile-time constant (0 LM, 1 traditional dogleg) -- the device kernels are
// instantiated per strategy so that the code of the other one does not weigh on the register allocation.
template <int STRAT>
EA_HD inline bool lm_strategy_step(LMState *s, LMCold *c, const LMOptions *o, const double As[21],
                                   const double gs[6], double step[6]) {
  if constexpr (STRAT == 0) {
    if (!s->reuse_diagonal) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        s->diagonal[i] = fmin(fmax(As[sym6(i, i)], o->min_lm_diagonal), o->max_lm_diagonal);
    }
    // Ceres hands the solver D = sqrt(diagonal / radius) and the solver squares it again; D^2 is used directly
    double D2[6], y[6];
    const double inv_radius = ea_rcp(s->radius);
#pragma unroll
    for (int i = 0; i < 6; ++i) D2[i] = s->diagonal[i] * inv_radius;
    s->reuse_diagonal = 1;
    if (EA_UNLIKELY(!solve_spd6(As, D2, gs, y))) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) step[i] = -y[i];
    return true;
  } else {
  // traditional dogleg
  if (!s->dl_reuse) {
    double diag[6], grad[6], gn[6];
    for (int i = 0; i < 6; ++i)
      diag[i] = sqrt(fmin(fmax(As[sym6(i, i)], o->min_lm_diagonal), o->max_lm_diagonal));
    for (int i = 0; i < 6; ++i) grad[i] = gs[i] / diag[i];
    double v[6], qf = 0.0, g2 = 0.0;
    for (int i = 0; i < 6; ++i) v[i] = grad[i] / diag[i];
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) qf += v[a] * As[sym6(a, b)] * v[b];
    for (int i = 0; i < 6; ++i) g2 += grad[i] * grad[i];
    s->alpha = g2 / qf;
    bool ok = false;
    while (s->mu < 1.0) {
      double D2[6], y[6];
      for (int i = 0; i < 6; ++i) D2[i] = diag[i] * diag[i] * s->mu;
      ok = solve_spd6(As, D2, gs, y);
      if (ok) {
        for (int i = 0; i < 6; ++i) gn[i] = y[i];
        break;
      }
      s->mu *= 10.0;
    }
    if (!ok) return false;
    s->mu = fmax(1e-8, 2.0 * s->mu / 10.0);
    for (int i = 0; i < 6; ++i) gn[i] *= -diag[i];
    for (int i = 0; i < 6; ++i) { c->dl_diag[i] = diag[i]; c->dl_grad[i] = grad[i]; c->dl_gn[i] = gn[i]; }
  }
  double dl_diag[6], dl_grad[6], dl_gn[6];
  for (int i = 0; i < 6; ++i) { dl_diag[i] = c->dl_diag[i]; dl_grad[i] = c->dl_grad[i]; dl_gn[i] = c->dl_gn[i]; }
  const double gn_norm = norm_n(dl_gn, 6);
  if (gn_norm <= s->radius) {
    for (int i = 0; i < 6; ++i) step[i] = dl_gn[i] / dl_diag[i];
    s->dogleg_step_norm = gn_norm;
    return true;
  }
  const double gradient_norm = norm_n(dl_grad, 6);
  if (gradient_norm * s->alpha >= s->radius) {
    for (int i = 0; i < 6; ++i)
      step[i] = -(s->radius / gradient_norm) * dl_grad[i] / dl_diag[i];
    s->dogleg_step_norm = s->radius;
    return true;
  }
  double b_dot_a = 0.0;
  for (int i = 0; i < 6; ++i) b_dot_a += -s->alpha * dl_grad[i] * dl_gn[i];
  const double a_sq = (s->alpha * gradient_norm) * (s->alpha * gradient_norm);
  const double bma_sq = a_sq - 2.0 * b_dot_a + gn_norm * gn_norm;
  const double cc = b_dot_a - a_sq;
  const double d = sqrt(cc * cc + bma_sq * (s->radius * s->radius - a_sq));
  const double beta = (cc <= 0) ? (d - cc) / bma_sq : (s->radius * s->radius - a_sq) / (d + cc);
  double dl[6];
  for (int i = 0; i < 6; ++i) dl[i] = (-s->alpha * (1.0 - beta)) * dl_grad[i] + beta * dl_gn[i];
  s->dogleg_step_norm = norm_n(dl, 6);
  for (int i = 0; i < 6; ++i) step[i] = dl[i] / dl_diag[i];
  return true;
  }
}

// Top of TrustRegionMinimizer's loop: convergence checks, then a trust-region step and the
// candidate pose.  Loops over invalid steps (they need no new evaluation).  On return either
// s->running == 0 or s->cand holds the pose to evaluate next.
// `fresh`: acc holds the evaluation at s->x (accepted step, first iteration) and is used from registers; otherwise
// the system is re-read from the cold state (after a rejected step).
template <int STRAT, bool LITE = false>
EA_HD inline void lm_prepare_next(LMState *s, LMCold *c, LMTrace *tr, const LMOptions *o,
                                  const double acc[kAccSlots], bool fresh) {
  double As[21], gs[6];
  bool have_system = false;
  for (;;) {
    if (EA_UNLIKELY(s->iteration >= o->max_num_iterations)) { lm_finish(s, 1, 4); return; }
    if constexpr (!LITE)
      if (EA_UNLIKELY(s->gradient_max_norm <= o->gradient_tolerance)) { lm_finish(s, 0, 2); return; }
    if (EA_UNLIKELY(s->radius <= o->min_trust_region_radius)) { lm_finish(s, 0, 5); return; }
    s->iteration += 1;
    if (EA_LIKELY(!have_system)) {
      double A[21], g[6];
      if (EA_LIKELY(fresh)) {
#pragma unroll
        for (int k = 0; k < 21; ++k) A[k] = acc[kAccJtJ + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = acc[kAccJtr + k];
      } else {
#pragma unroll
        for (int k = 0; k < 21; ++k) A[k] = c->A[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = c->g[k];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        gs[a] = g[a] * s->S[a];
#pragma unroll
        for (int b = a; b < 6; ++b) As[sym6(a, b)] = A[sym6(a, b)] * s->S[a] * s->S[b];
      }
      have_system = true;
    }
    EA_LM_PROBE(2);
    double step[6];
    bool ok = lm_strategy_step<STRAT>(s, c, o, As, gs, step);
    EA_LM_PROBE(3);
    if (EA_LIKELY(ok)) {
      // model_cost_change = -(Js s)^T (r + Js s / 2) = -(g^T s + s^T A s / 2)
      double gts = 0.0, diag = 0.0, off = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        gts += gs[a] * step[a];
        diag += step[a] * As[sym6(a, a)] * step[a];
#pragma unroll
        for (int b = a + 1; b < 6; ++b) off += step[a] * As[sym6(a, b)] * step[b];
      }
      s->model_cost_change = -(gts + 0.5 * (diag + 2.0 * off));
      if (!(s->model_cost_change > 0.0)) ok = false;
    }
    EA_LM_PROBE(4);
    if (EA_LIKELY(ok)) {
      double delta[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) delta[i] = step[i] * s->S[i];
      pose_plus(s->x, delta, s->cand);
      s->num_consecutive_invalid = 0;
      EA_LM_PROBE(5);
      return;
    }
    // HandleInvalidStep
    s->num_unsuccessful += 1;
    lm_trace(s, tr, s->iteration, 0.0, 0.0, 0.0, 0);
    if (++s->num_consecutive_invalid >= o->max_num_consecutive_invalid_steps) {
      lm_finish(s, 2, 7);
      return;
    }
    if constexpr (STRAT == 0) { s->radius *= 0.5; s->reuse_diagonal = 1; }
    else { s->mu *= 10.0; s->dl_reuse = 0; }
  }
}

// An evaluation is usable when no functor returned false AND cost, JtJ and Jtr are all finite -- ceres:
// ResidualBlock::Evaluate's IsArrayValid check fails an evaluation whose residuals or Jacobians hold a NaN / Inf.  A finite
// cost does not imply a finite system (fp32 products can overflow where the residual itself does not), and a non-finite
// system would go through the factorisation and publish a NaN candidate pose.  The sum of the magnitudes of the 28 slots
// is finite exactly when every slot is; four independent chains keep it off the lane's dependency chain.
EA_HD inline bool lm_eval_usable(const double acc[kAccSlots]) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int i = 0; i < 28; i += 4) {
    s0 += fabs(acc[i]); s1 += fabs(acc[i + 1]); s2 += fabs(acc[i + 2]); s3 += fabs(acc[i + 3]);
  }
  static_assert(kAccCost == 27 && kAccInvalid == 28, "slots 0..27 = JtJ, Jtr, cost");
  return !(acc[kAccInvalid] > 0.0) && ((s0 + s1) + (s2 + s3) <= DBL_MAX);
}

// after the evaluation at the initial pose
template <int STRAT, bool LITE = false>
EA_HD inline void lm_begin(LMState *s, LMCold *c, LMTrace *tr, const LMOptions *o, const double acc[kAccSlots],
                           LMPending *pend) {
  pend->store_system = 0;
  pend->trace_it = -1;
  s->num_evals += 1;
  // a functor returning false, or a non-finite residual / Jacobian (lm_eval_usable), fails the evaluation: at the start
  // point the solve ends with FAILURE and the parameters untouched
  if (EA_UNLIKELY(!lm_eval_usable(acc))) { lm_finish(s, 2, 6); return; }
  lm_take_system<LITE>(s, pend, acc);
  if (o->jacobi_scaling)
    for (int i = 0; i < 6; ++i) s->S[i] = 1.0 / (1.0 + sqrt(acc[kAccJtJ + sym6(i, i)]));
  lm_pend_trace(s, pend, 0, 0.0, 0.0, 0.0, 1);
  lm_prepare_next<STRAT, LITE>(s, c, tr, o, acc, true);
}

// The usual iteration of the LM strategy as ONE straight line: usable evaluation, no convergence test fires, the step is
// accepted, the factorisation succeeds and the model cost decreases.  It is lm_advance's own arithmetic, statement for
// statement and in the same order (so both give the same bits), minus the joins: in the general form every variable that the
// accept and the reject branch, the fresh and the stored system, the valid and the invalid step set differently is copied at
// the join, and on the one lane that runs this code a register move costs what an FMA costs -- roughly a third of the
// instructions of an iteration were moves.  Works on locals and commits at the end: on anything unusual it returns false with
// *s and *pend untouched, and the caller runs the general form.
template <int STRAT, bool LITE>
EA_HD inline bool lm_advance_fast(LMState *s, LMCold *c, const LMOptions *o, const double acc[kAccSlots], LMPending *pend) {
  if (EA_UNLIKELY(!lm_eval_usable(acc))) return false;
  const double cand_cost = acc[kAccCost];
  double dx[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) dx[i] = s->x[i] - s->cand[i];
  const double step_norm = norm_n(dx, 7);
  if (EA_UNLIKELY(step_norm <= o->parameter_tolerance * (s->x_norm + o->parameter_tolerance))) return false;
  const double cost_change = s->cost - cand_cost;
  if (EA_UNLIKELY(fabs(cost_change) <= o->function_tolerance * s->cost)) return false;
  const double rel = cost_change / s->model_cost_change;
  if (EA_UNLIKELY(!(rel > o->min_relative_decrease))) return false;
  // accepted: the system of this evaluation is the system at the new x
  double x_norm = s->x_norm, gradient_max_norm = s->gradient_max_norm;
  if constexpr (!LITE) {
    x_norm = norm_n(s->cand, 7);
    double neg[6], xp[7], m = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) neg[i] = -acc[kAccJtr + i];
    pose_plus(s->cand, neg, xp, /*small_expected=*/false);
#pragma unroll
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(s->cand[i] - xp[i]));
    gradient_max_norm = m;
  }
  double radius = s->radius;
  if constexpr (STRAT == 0) {
    const double f = 2.0 * rel - 1.0;
    radius = s->radius / fmax(1.0 / 3.0, 1.0 - f * f * f);
    radius = fmin(o->max_trust_region_radius, radius);
  } else {
    if (rel < 0.25) radius *= 0.5;
    if (rel > 0.75) radius = fmax(radius, 3.0 * s->dogleg_step_norm);
    radius = fmin(radius, o->max_trust_region_radius);
  }
  // top of the next iteration
  if (EA_UNLIKELY(s->iteration >= o->max_num_iterations)) return false;
  if constexpr (!LITE)
    if (EA_UNLIKELY(gradient_max_norm <= o->gradient_tolerance)) return false;
  if (EA_UNLIKELY(radius <= o->min_trust_region_radius)) return false;
  double As[21], gs[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    gs[a] = acc[kAccJtr + a] * s->S[a];
#pragma unroll
    for (int b = a; b < 6; ++b) As[sym6(a, b)] = acc[kAccJtJ + sym6(a, b)] * s->S[a] * s->S[b];
  }
  double diagonal[6], D2[6], y[6], step[6];
  // (dogleg) what lm_strategy_step<1> keeps for the steps after a rejected one, and the scalars it moves
  double dl_diag[6], dl_grad[6], dl_gn[6], mu = s->mu, alpha = s->alpha, dogleg_step_norm = s->dogleg_step_norm;
  if constexpr (STRAT == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) diagonal[i] = fmin(fmax(As[sym6(i, i)], o->min_lm_diagonal), o->max_lm_diagonal);
    const double inv_radius = ea_rcp(radius);
#pragma unroll
    for (int i = 0; i < 6; ++i) D2[i] = diagonal[i] * inv_radius;
    if (EA_UNLIKELY(!solve_spd6(As, D2, gs, y))) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) step[i] = -y[i];
  } else {
    // traditional dogleg after an accepted step (dl_reuse == 0): lm_strategy_step<1>'s statements, the Gauss-Newton solve at
    // the first mu only (a failed factorisation goes to the general form and its loop over mu)
#pragma unroll
    for (int i = 0; i < 6; ++i) dl_diag[i] = sqrt(fmin(fmax(As[sym6(i, i)], o->min_lm_diagonal), o->max_lm_diagonal));
#pragma unroll
    for (int i = 0; i < 6; ++i) dl_grad[i] = gs[i] / dl_diag[i];
    double v[6], qf = 0.0, g2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = dl_grad[i] / dl_diag[i];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) qf += v[a] * As[sym6(a, b)] * v[b];
#pragma unroll
    for (int i = 0; i < 6; ++i) g2 += dl_grad[i] * dl_grad[i];
    alpha = g2 / qf;
    if (EA_UNLIKELY(!(mu < 1.0))) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) D2[i] = dl_diag[i] * dl_diag[i] * mu;
    if (EA_UNLIKELY(!solve_spd6(As, D2, gs, y))) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) dl_gn[i] = y[i];
    mu = fmax(1e-8, 2.0 * mu / 10.0);
#pragma unroll
    for (int i = 0; i < 6; ++i) dl_gn[i] *= -dl_diag[i];
    const double gn_norm = norm_n(dl_gn, 6);
    if (gn_norm <= radius) {
#pragma unroll
      for (int i = 0; i < 6; ++i) step[i] = dl_gn[i] / dl_diag[i];
      dogleg_step_norm = gn_norm;
    } else {
      const double gradient_norm = norm_n(dl_grad, 6);
      if (gradient_norm * alpha >= radius) {
#pragma unroll
        for (int i = 0; i < 6; ++i) step[i] = -(radius / gradient_norm) * dl_grad[i] / dl_diag[i];
        dogleg_step_norm = radius;
      } else {
        double b_dot_a = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) b_dot_a += -alpha * dl_grad[i] * dl_gn[i];
        const double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
        const double bma_sq = a_sq - 2.0 * b_dot_a + gn_norm * gn_norm;
        const double cc = b_dot_a - a_sq;
        const double d = sqrt(cc * cc + bma_sq * (radius * radius - a_sq));
        const double beta = (cc <= 0) ? (d - cc) / bma_sq : (radius * radius - a_sq) / (d + cc);
        double dl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) dl[i] = (-alpha * (1.0 - beta)) * dl_grad[i] + beta * dl_gn[i];
        dogleg_step_norm = norm_n(dl, 6);
#pragma unroll
        for (int i = 0; i < 6; ++i) step[i] = dl[i] / dl_diag[i];
      }
    }
  }
  double gts = 0.0, diag = 0.0, off = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    gts += gs[a] * step[a];
    diag += step[a] * As[sym6(a, a)] * step[a];
#pragma unroll
    for (int b = a + 1; b < 6; ++b) off += step[a] * As[sym6(a, b)] * step[b];
  }
  const double model_cost_change = -(gts + 0.5 * (diag + 2.0 * off));
  if (EA_UNLIKELY(!(model_cost_change > 0.0))) return false;
  double delta[6], cand[7];
#pragma unroll
  for (int i = 0; i < 6; ++i) delta[i] = step[i] * s->S[i];
  pose_plus(s->cand, delta, cand);
  // ---- commit (the order of lm_advance: trace row of the accepted step, then the new iteration's state)
  pend->store_system = 1;
#pragma unroll
  for (int i = 0; i < 7; ++i) s->x[i] = s->cand[i];
  s->x_norm = x_norm;
  s->cost = cand_cost;
  s->gradient_max_norm = gradient_max_norm;
  s->num_evals += 1;
  s->num_successful += 1;
  s->radius = radius;
  if constexpr (STRAT == 0) s->decrease_factor = 2.0;
  lm_pend_trace(s, pend, s->iteration, cost_change, step_norm, rel, 1);
  s->iteration += 1;
  if constexpr (STRAT == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) s->diagonal[i] = diagonal[i];
    s->reuse_diagonal = 1;
  } else {
    s->mu = mu; s->alpha = alpha; s->dogleg_step_norm = dogleg_step_norm;
    s->dl_reuse = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { c->dl_diag[i] = dl_diag[i]; c->dl_grad[i] = dl_grad[i]; c->dl_gn[i] = dl_gn[i]; }
  }
  s->model_cost_change = model_cost_change;
#pragma unroll
  for (int i = 0; i < 7; ++i) s->cand[i] = cand[i];
  s->num_consecutive_invalid = 0;
  return true;
}

// after the evaluation at s->cand
template <int STRAT, bool LITE = false>
EA_HD inline void lm_advance(LMState *s, LMCold *c, LMTrace *tr, const LMOptions *o, const double acc[kAccSlots],
                             LMPending *pend) {
  pend->store_system = 0;
  pend->trace_it = -1;
#ifndef EA_LM_NO_FAST_PATH  // (tests/test_lm_host_logic.py builds the host shim both ways: the two must agree bit for bit)
  if (EA_LIKELY((lm_advance_fast<STRAT, LITE>(s, c, o, acc, pend)))) return;
#endif
  s->num_evals += 1;
  const bool eval_ok = lm_eval_usable(acc);  // (false: the step is rejected like one that raised the cost)
  const double cand_cost = eval_ok ? acc[kAccCost] : DBL_MAX;
  double dx[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) dx[i] = s->x[i] - s->cand[i];
  const double step_norm = norm_n(dx, 7);
  if (EA_UNLIKELY(step_norm <= o->parameter_tolerance * (s->x_norm + o->parameter_tolerance))) {
    lm_pend_trace(s, pend, s->iteration, 0.0, step_norm, 0.0, 0);
    lm_finish(s, 0, 3);
    return;
  }
  const double cost_change = s->cost - cand_cost;
  if (EA_UNLIKELY(fabs(cost_change) <= o->function_tolerance * s->cost)) {
    lm_pend_trace(s, pend, s->iteration, cost_change, step_norm, 0.0, 0);
    lm_finish(s, 0, 1);
    return;
  }
  const double rel = cost_change / s->model_cost_change;
  EA_LM_PROBE(0);
  bool fresh = false;
  if (EA_LIKELY(rel > o->min_relative_decrease)) {
#pragma unroll
    for (int i = 0; i < 7; ++i) s->x[i] = s->cand[i];
    if constexpr (!LITE) s->x_norm = norm_n(s->x, 7);
    lm_take_system<LITE>(s, pend, acc);
    fresh = true;
    EA_LM_PROBE(1);
    s->num_successful += 1;
    if constexpr (STRAT == 0) {
      const double f = 2.0 * rel - 1.0;
      s->radius = s->radius / fmax(1.0 / 3.0, 1.0 - f * f * f);
      s->radius = fmin(o->max_trust_region_radius, s->radius);
      s->decrease_factor = 2.0;
      s->reuse_diagonal = 0;
    } else {
      if (rel < 0.25) s->radius *= 0.5;
      if (rel > 0.75) s->radius = fmax(s->radius, 3.0 * s->dogleg_step_norm);
      s->radius = fmin(s->radius, o->max_trust_region_radius);
      s->dl_reuse = 0;
    }
    EA_LM_PROBE(6);
    lm_pend_trace(s, pend, s->iteration, cost_change, step_norm, rel, 1);
  } else {
    s->num_unsuccessful += 1;
    if constexpr (STRAT == 0) {
      s->radius = s->radius / s->decrease_factor;
      s->decrease_factor *= 2.0;
      s->reuse_diagonal = 1;
    } else {
      s->radius *= 0.5;
      s->dl_reuse = 1;
    }
    lm_pend_trace(s, pend, s->iteration, cost_change, step_norm, rel, 0);
  }
  EA_LM_PROBE(7);
  lm_prepare_next<STRAT, LITE>(s, c, tr, o, acc, fresh);
}

// run-time strategy (host shim)
EA_HD inline void lm_begin_rt(LMState *s, LMCold *c, LMTrace *tr, const LMOptions *o, const double acc[kAccSlots],
                              LMPending *pend) {
  if (o->strategy == 0) lm_begin<0>(s, c, tr, o, acc, pend); else lm_begin<1>(s, c, tr, o, acc, pend);
}
EA_HD inline void lm_advance_rt(LMState *s, LMCold *c, LMTrace *tr, const LMOptions *o, const double acc[kAccSlots],
                                LMPending *pend) {
  if (o->strategy == 0) lm_advance<0>(s, c, tr, o, acc, pend); else lm_advance<1>(s, c, tr, o, acc, pend);
}

}  // namespace ea

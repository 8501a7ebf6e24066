/*
 * ea_oracle.h — CPU restatement (fp64, plain C) of the edge-alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under edge_alignment_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED.  The reference (kuwt/edge_alignment) has no tests, no golden vectors and
 * cannot be compiled here: its arithmetic lives in Ceres Solver (not vendored; the author's
 * log shows 1.12.0, `standalone/README.md:28`; the API used exists in 1.12 … 2.1), Eigen 3 and
 * OpenCV 3, none of which are in this image.  This file therefore restates
 *   - the reference's own functor text           standalone/utils.h:48-80
 *   - its problem set-up / call pattern           standalone/standalone_edge_align.cpp:256-300
 *   - its pose in/out convention (w,x,y,z | t)    standalone/PoseManipUtils.cpp:3-27
 * and the *published* Ceres algorithms it calls (cubic_interpolation.h, jet.h,
 * local_parameterization.cc, loss_function.cc, corrector.cc, trust_region_minimizer.cc,
 * levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc, dense_qr_solver.cc).
 * What pins it: closed-form known answers (polynomial reproduction of the Catmull-Rom
 * spline, texel reproduction at integer coordinates, clamp addressing), the agreement of two
 * independent Jacobian derivations (forward-mode Jet<7> of the literal functor  vs  analytic
 * 1x6 row) plus finite differences and a numpy restatement (oracle/ea_numpy.py), and planted-
 * pose recoveries.  The only reference-held number reproduced is the residual-block count
 * 1482 = ceil(44457/30) of `standalone/README.md:34` (tests/test_golden_and_preprocess.py).
 */
#ifndef EA_ORACLE_H
#define EA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { EA_ORACLE_LOSS_TRIVIAL = 0, EA_ORACLE_LOSS_CAUCHY = 1, EA_ORACLE_LOSS_HUBER = 2 };
enum { EA_ORACLE_JAC_ANALYTIC = 0, EA_ORACLE_JAC_JET = 1 };
enum { EA_ORACLE_LIN_CHOLESKY = 0, EA_ORACLE_LIN_DENSE_QR = 1 };
enum { EA_ORACLE_STRATEGY_LM = 0, EA_ORACLE_STRATEGY_DOGLEG = 1 };

/* termination codes follow ceres::TerminationType's meaning */
enum {
  EA_ORACLE_CONVERGENCE = 0,
  EA_ORACLE_NO_CONVERGENCE = 1,
  EA_ORACLE_FAILURE = 2
};
/* why (finer than Ceres' enum; mirrors the message it prints) */
enum {
  EA_ORACLE_WHY_NONE = 0,
  EA_ORACLE_WHY_FUNCTION_TOL = 1,
  EA_ORACLE_WHY_GRADIENT_TOL = 2,
  EA_ORACLE_WHY_PARAMETER_TOL = 3,
  EA_ORACLE_WHY_MAX_ITERATIONS = 4,
  EA_ORACLE_WHY_MIN_RADIUS = 5,
  EA_ORACLE_WHY_INITIAL_EVAL_FAILED = 6,
  EA_ORACLE_WHY_TOO_MANY_INVALID_STEPS = 7,
  EA_ORACLE_WHY_EVAL_FAILED = 8
};

typedef struct {
  double fx, fy, cx, cy;
  /* ceres::Grid2D<double,1> view exactly as the reference constructs it
   * (standalone_edge_align.cpp:258): row-major, value(r,c) = grid[r*grid_cols + c];
   * the functor calls Evaluate(r = u, c = v)  (utils.h:77). */
  const double *grid;
  int grid_rows, grid_cols;
  int loss_kind;   /* EA_ORACLE_LOSS_* ; the reference passes `new CauchyLoss(1.)` (:272) */
  double loss_a;
  /* functor flavour knobs; standalone = {0.01, 0.0, 0} */
  double z_guard;  /* evaluation fails if -z_guard < bz < z_guard   (utils.h:70-73); 0 = no guard */
  double z_eps;    /* divisor is bz + z_eps                         (EAResidue.h:104-105) */
  int rot_transposed; /* apply R^T (EAResidue.h:90,99-101) */
  /* residual variants of standalone/utils.h */
  int use_distortion;      /* EAResidueEx / EAResidueSecondCamEx (utils.h:102-177, :295-421) */
  double k1, k2, p1, p2, k3;
  int use_second_cam;      /* EAResidueSecondCam[Ex] (utils.h:179-292): b_T_a_SecCam = T12 * b_T_a * T12inv */
  double T12[16], T12inv[16]; /* row-major 4x4, as the ptrans_1to2 / ptrans_1to2_inv arrays */
} ea_oracle_problem;

/* one residual family of a problem: a functor flavour + its points; several terms share (q,t)
 * (standalone_edge_align.cpp:791-803: EAResidue blocks of camera 1 + EAResidueSecondCam blocks of camera 2) */
typedef struct {
  const ea_oracle_problem *problem;
  const double *xyz;
  int64_t n;
  int stride;
} ea_oracle_term;

#define EA_ORACLE_MAX_ITERS 512

typedef struct {
  int max_num_iterations;        /* 50  */
  double function_tolerance;     /* 1e-6 */
  double gradient_tolerance;     /* 1e-10 */
  double parameter_tolerance;    /* 1e-8 */
  double initial_trust_region_radius; /* 1e4 */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double min_lm_diagonal;             /* 1e-6 */
  double max_lm_diagonal;             /* 1e32 */
  int max_num_consecutive_invalid_steps; /* 5 */
  int jacobi_scaling;                 /* 1 */
  int jacobian_mode;                  /* EA_ORACLE_JAC_* */
  int linear_solver;                  /* EA_ORACLE_LIN_* */
  int strategy;                       /* EA_ORACLE_STRATEGY_* */
  int verbose;
} ea_oracle_options;

typedef struct {
  int termination;      /* EA_ORACLE_CONVERGENCE / NO_CONVERGENCE / FAILURE */
  int why;
  int num_iterations;   /* index of the last recorded iteration (iteration 0 = initial eval) */
  int num_successful_steps, num_unsuccessful_steps;
  double initial_cost, final_cost;
  int64_t num_residual_evals, num_jacobian_evals; /* point-evals */
  /* per-iteration trace, entries [0 .. num_iterations] */
  double it_cost[EA_ORACLE_MAX_ITERS];
  double it_cost_change[EA_ORACLE_MAX_ITERS];
  double it_gradient_max_norm[EA_ORACLE_MAX_ITERS];
  double it_step_norm[EA_ORACLE_MAX_ITERS];
  double it_relative_decrease[EA_ORACLE_MAX_ITERS];
  double it_radius[EA_ORACLE_MAX_ITERS];
  int it_successful[EA_ORACLE_MAX_ITERS];
} ea_oracle_summary;

void ea_oracle_default_options(ea_oracle_options *o);
void ea_oracle_default_problem(ea_oracle_problem *p);

/* ceres::BiCubicInterpolator<Grid2D<double,1>>::Evaluate(r, c, f, dfdr, dfdc) */
void ea_oracle_bicubic(const double *grid, int rows, int cols, double r, double c,
                       double *f, double *dfdr, double *dfdc);

/* One residual block the way Ceres' AutoDiffCostFunction<EAResidue,1,4,3> evaluates it:
 * Jet<double,7> through the literal functor.  Returns 1 on success, 0 if the functor
 * returned false.  jq = d r / d(w,x,y,z), jt = d r / d t.  No loss applied. */
int ea_oracle_block_jet(const ea_oracle_problem *p, const double q[4], const double t[3],
                        const double X[3], double *r, double jq[4], double jt[3]);

/* Same block, analytic: r and the 1x6 row in Ceres' reduced ordering [delta(3) | t(3)].
 * No loss applied. */
int ea_oracle_block_analytic(const ea_oracle_problem *p, const double q[4], const double t[3],
                             const double X[3], double *r, double j6[6]);

/* QuaternionParameterization::Plus and ::ComputeJacobian (4x3 row-major) */
void ea_oracle_quat_plus(const double q[4], const double delta[3], double q_plus[4]);
void ea_oracle_quat_plus_jacobian(const double q[4], double P[12]);

/* Whole-problem evaluation = what Ceres' evaluator hands the minimiser:
 *   r_out[n]   loss-corrected residuals (sqrt(rho') r)      (nullable)
 *   J_out[n*6] loss-corrected local Jacobian rows            (nullable)
 *   raw_r[n], raw_J[n*6]  uncorrected                        (nullable)
 *   cost = 1/2 sum rho(r^2);  JtJ[36] row-major symmetric;  Jtr[6] (= gradient)
 * xyz: n points, AoS with `stride` doubles between points (3 or 4).
 * Returns number of blocks whose evaluation failed (Ceres: any failure fails the whole
 * evaluation); sums cover the valid blocks only. */
int64_t ea_oracle_eval(const ea_oracle_problem *p, const double *xyz, int64_t n, int stride,
                       const double q[4], const double t[3], int jacobian_mode,
                       double *cost, double JtJ[36], double Jtr[6],
                       double *r_out, double *J_out, double *raw_r, double *raw_J);

/* residual-only evaluation (candidate cost) */
int64_t ea_oracle_cost(const ea_oracle_problem *p, const double *xyz, int64_t n, int stride,
                       const double q[4], const double t[3], double *cost);

/* multi-term versions: sums over all terms (one ceres::Problem with several residual families) */
int64_t ea_oracle_eval_terms(const ea_oracle_term *terms, int nterms, const double q[4], const double t[3],
                             int jacobian_mode, double *cost, double JtJ[36], double Jtr[6]);
int ea_oracle_solve_terms(const ea_oracle_term *terms, int nterms, const ea_oracle_options *opt, double q[4],
                          double t[3], ea_oracle_summary *summary);

/* ceres::Solve(options, &problem, &summary) for this problem shape; q,t updated in place */
int ea_oracle_solve(const ea_oracle_problem *p, const double *xyz, int64_t n, int stride,
                    const ea_oracle_options *opt, double q[4], double t[3],
                    ea_oracle_summary *summary);

#ifdef __cplusplus
}
#endif
#endif

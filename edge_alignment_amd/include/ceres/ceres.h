// ceres/ceres.h — the slice of the ceres:: API that kuwt/edge_alignment touches on its hot path,
// re-hosted on the MI355X C-ABI (include/ea_hip.h).  NOT Ceres Solver and not a general solver:
// a source-compatibility facade so that the reference's "Setup non-linear Least Squares" blocks
// (standalone/standalone_edge_align.cpp:256-293, src/SolveEA.cpp:124-216) compile and run
// unchanged against libea_hip.so.
//
// What it understands: residual blocks created by EAResidue::Create(...) /
// AutoDiffCostFunction<EAResidue,1,4,3> over ONE interpolator and ONE (q,t) pair, an optional
// CauchyLoss / HuberLoss / TrivialLoss (NULL = trivial), QuaternionParameterization on q, and the
// Solver::Options fields the reference sets.  Anything else is reported through
// Summary::termination_type = FAILURE with a message; nothing is evaluated on the CPU.
#pragma once

#include <cmath>
#include <cstdio>
#include <memory>
#include <sstream>
#include <string>
#include <algorithm>
#include <unordered_set>
#include <vector>

#include "../../../include/ea_hip.h"
#include "cubic_interpolation.h"
#include "jet.h"
#include "loss_function.h"
#include "rotation.h"

namespace ceres {

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum MinimizerType { LINE_SEARCH, TRUST_REGION };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum DoglegType { TRADITIONAL_DOGLEG, SUBSPACE_DOGLEG };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

// ---- cost functions ------------------------------------------------------------------------
// The facade never differentiates anything: a cost function is only a carrier of the per-block
// constants (3-D point, intrinsics, interpolator) that the GPU kernels consume.
struct EABlockInfo {
  double fx, fy, cx, cy;
  double X, Y, Z;
  const double *grid_data;  // Grid2D view the functor samples at (r = u, c = v)
  int grid_rows, grid_cols;
  double z_guard, z_eps;    // functor flavour (standalone: 0.01, 0; ROS: 0, 0.001)
  int rot_transposed;
  // residual variants (utils.h:102-421); zero-initialised = plain EAResidue
  int variant = 0;          // bit 0: distortion (k1,k2,p1,p2,k3), bit 1: second camera (T12, T12inv)
  double dist[5] = {0, 0, 0, 0, 0};
  double T12[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double T12inv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

class CostFunction {
 public:
  virtual ~CostFunction() {}
  // Ceres' interface: residuals and (nullable, per parameter block nullable) row-major Jacobians of ONE block on the host.
  // A probe for spot checks -- ceres::Solve never calls it, every block is evaluated by the gfx950 kernels.
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const {
    (void)parameters; (void)residuals; (void)jacobians;
    return false;
  }
  // true if this block is an edge-alignment block; fills info
  virtual bool DescribeEdgeAlignmentBlock(EABlockInfo *) const { return false; }
};

// AutoDiffCostFunction<Functor, 1, 4, 3>: owns the functor like Ceres does.  The functor must
// expose `bool ea_describe(ceres::EABlockInfo*) const` (edge_alignment_amd/include/EAResidue.h does).
template <typename Functor, int kNumResiduals, int N0 = 0, int N1 = 0, int N2 = 0>
class AutoDiffCostFunction : public CostFunction {
 public:
  explicit AutoDiffCostFunction(Functor *functor) : functor_(functor) {}
  // forward-mode differentiation of the functor's templated operator() with Jet<double, N0 + N1> (two parameter
  // blocks, as every cost function of the reference has: q[4], t[3])
  bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
    static_assert(N2 == 0, "two parameter blocks");
    if (!jacobians) return (*functor_)(parameters[0], parameters[1], residuals);
    typedef Jet<double, N0 + N1> J;
    J p0[N0 > 0 ? N0 : 1], p1[N1 > 0 ? N1 : 1], r[kNumResiduals];
    for (int i = 0; i < N0; ++i) p0[i] = J(parameters[0][i], i);
    for (int i = 0; i < N1; ++i) p1[i] = J(parameters[1][i], N0 + i);
    if (!(*functor_)(p0, p1, r)) return false;
    for (int k = 0; k < kNumResiduals; ++k) {
      residuals[k] = r[k].a;
      if (jacobians[0]) for (int i = 0; i < N0; ++i) jacobians[0][k * N0 + i] = r[k].v[i];
      if (jacobians[1]) for (int i = 0; i < N1; ++i) jacobians[1][k * N1 + i] = r[k].v[N0 + i];
    }
    return true;
  }
  bool DescribeEdgeAlignmentBlock(EABlockInfo *info) const override {
    if (kNumResiduals != 1 || N0 != 4 || N1 != 3 || N2 != 0) return false;
    return functor_->ea_describe(info);
  }
  const Functor *functor() const { return functor_.get(); }

 private:
  std::unique_ptr<Functor> functor_;
};

// ---- local parameterisation -----------------------------------------------------------------
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;  // GlobalSize x LocalSize, row-major
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
  virtual bool IsQuaternion() const { return false; }
};
// q+ = [cos|d|, sin|d|/|d| d] (x) q  — applied on the device by the LM-step kernel; the host forms below are Ceres'
// public definition (local_parameterization.cc), for callers that probe the parameterisation themselves
class QuaternionParameterization : public LocalParameterization {
 public:
  bool Plus(const double *x, const double *delta, double *x_plus_delta) const override {
    const double n = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (n > 0.0) {
      const double s = std::sin(n) / n;
      const double d[4] = {std::cos(n), s * delta[0], s * delta[1], s * delta[2]};
      x_plus_delta[0] = d[0] * x[0] - d[1] * x[1] - d[2] * x[2] - d[3] * x[3];
      x_plus_delta[1] = d[0] * x[1] + d[1] * x[0] + d[2] * x[3] - d[3] * x[2];
      x_plus_delta[2] = d[0] * x[2] - d[1] * x[3] + d[2] * x[0] + d[3] * x[1];
      x_plus_delta[3] = d[0] * x[3] + d[1] * x[2] - d[2] * x[1] + d[3] * x[0];
    } else {
      for (int i = 0; i < 4; ++i) x_plus_delta[i] = x[i];
    }
    return true;
  }
  bool ComputeJacobian(const double *x, double *J) const override {
    J[0] = -x[1]; J[1] = -x[2]; J[2] = -x[3];
    J[3] = x[0];  J[4] = x[3];  J[5] = -x[2];
    J[6] = -x[3]; J[7] = x[0];  J[8] = x[1];
    J[9] = x[2];  J[10] = -x[1]; J[11] = x[0];
    return true;
  }
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 3; }
  bool IsQuaternion() const override { return true; }
};

// ---- problem --------------------------------------------------------------------------------
// ceres/crs_matrix.h: compressed-row sparse matrix as Problem::Evaluate hands it out
struct CRSMatrix {
  CRSMatrix() : num_rows(0), num_cols(0) {}
  int num_rows, num_cols;
  std::vector<int> cols, rows;
  std::vector<double> values;
};

class Problem {
 public:
  Problem() {}
  ~Problem() {
    for (auto *c : costs_) delete c;
    // a loss object may have been handed to several blocks (Ceres allows it): delete each once
    if (!std::is_sorted(losses_.begin(), losses_.end())) std::sort(losses_.begin(), losses_.end());
    losses_.erase(std::unique(losses_.begin(), losses_.end()), losses_.end());
    for (auto *l : losses_) delete l;
    for (auto *p : params_) delete p;
  }
  Problem(const Problem &) = delete;
  Problem &operator=(const Problem &) = delete;

  // standalone_edge_align.cpp:272  problem.AddResidualBlock(cost, new CauchyLoss(1.), q, t)
  void AddResidualBlock(CostFunction *cost, LossFunction *loss, double *q, double *t) {
    costs_.push_back(cost);
    if (loss && (losses_.empty() || losses_.back() != loss))
      losses_.push_back(loss);  // one `new CauchyLoss(1.)` per block in the reference; shared ones are fine too (~Problem)
    // What a block stores is its point and the index of its functor description: the reference adds one block per edge point
    // (44 457 per frame pair at stride 1), all describing the same functor; 64 bytes per block instead of the 430 of a full
    // description keep the list in cache while it is built and gathered (see profiles/r02_facade_timing.txt).
    EABlockInfo info;
    Block b;
    b.ok = cost->DescribeEdgeAlignmentBlock(&info);
    b.X = info.X; b.Y = info.Y; b.Z = info.Z;
    b.fam = -1;
    if (b.ok) {
      if (last_fam_ >= 0 && SameFunctor(fams_[(size_t)last_fam_], info)) b.fam = last_fam_;
      else {
        for (size_t k = 0; k < fams_.size() && b.fam < 0; ++k)
          if (SameFunctor(fams_[k], info)) b.fam = (int)k;
        if (b.fam < 0) { fams_.push_back(info); b.fam = (int)fams_.size() - 1; }
        last_fam_ = b.fam;
      }
    }
    b.loss = loss;
    b.q = q;
    b.t = t;
    blocks_.push_back(b);
  }
  // standalone_edge_align.cpp:277-278
  void SetParameterization(double *values, LocalParameterization *p) {
    params_.push_back(p);
    if (p && p->IsQuaternion()) quat_param_on_ = values;
  }
  int NumResidualBlocks() const { return (int)blocks_.size(); }
  int NumResiduals() const { return (int)blocks_.size(); }
  int NumParameterBlocks() const { return blocks_.empty() ? 0 : 2; }  // the quaternion and the translation
  int NumParameters() const { return blocks_.empty() ? 0 : 7; }

  // src/SolveEA.cpp:241  problem.Evaluate(Problem::EvaluateOptions(), &cost, &residuals, NULL, NULL)
  // cost = 1/2 sum rho(r^2); residuals: one per block in the order they were added, loss-corrected like Ceres'
  // apply_loss_function = true; gradient: the 6 tangent-space entries (J^T r); jacobian: one row per block in the same
  // order, six columns [d r / d delta (3) | d r / d t (3)] -- the parameter blocks' local sizes in the order
  // AddResidualBlock names them (quaternion, translation) -- as a compressed-row matrix with dense rows.
  struct EvaluateOptions {
    bool apply_loss_function = true;
    int ea_dtype = EA_F64;
    int ea_device = 0;
  };
  inline bool Evaluate(const EvaluateOptions &opt, double *cost, std::vector<double> *residuals,
                       std::vector<double> *gradient, CRSMatrix *jacobian);

 private:
  struct Block {
    double X, Y, Z;
    int fam;   // index into fams_: everything of the functor's description but the point
    bool ok;
    LossFunction *loss;
    double *q, *t;
  };
  static bool SameFunctor(const EABlockInfo &x, const EABlockInfo &y) {
    if (x.grid_data != y.grid_data || x.grid_rows != y.grid_rows || x.grid_cols != y.grid_cols) return false;
    if (x.fx != y.fx || x.fy != y.fy || x.cx != y.cx || x.cy != y.cy) return false;
    if (x.z_guard != y.z_guard || x.z_eps != y.z_eps || x.rot_transposed != y.rot_transposed) return false;
    if (x.variant != y.variant) return false;
    if (x.variant & 1)
      for (int i = 0; i < 5; ++i) if (x.dist[i] != y.dist[i]) return false;
    if (x.variant & 2)
      for (int i = 0; i < 16; ++i) if (x.T12[i] != y.T12[i] || x.T12inv[i] != y.T12inv[i]) return false;
    return true;
  }
  std::vector<Block> blocks_;
  std::vector<EABlockInfo> fams_;   // distinct functor descriptions (X, Y, Z of the entry unused)
  int last_fam_ = -1;
  std::vector<CostFunction *> costs_;
  std::vector<LossFunction *> losses_;
  std::vector<LocalParameterization *> params_;
  double *quat_param_on_ = nullptr;
  friend class ProblemAccess;
};

// ---- solver ---------------------------------------------------------------------------------
struct Solver {
  struct Options {
    MinimizerType minimizer_type = TRUST_REGION;
    TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
    DoglegType dogleg_type = TRADITIONAL_DOGLEG;
    LinearSolverType linear_solver_type = DENSE_QR;  // accepted; the 6x6 system is solved on the device
    int max_num_iterations = 50;
    double max_solver_time_in_seconds = 1e9;
    int num_threads = 1;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    int max_num_consecutive_invalid_steps = 5;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    bool jacobi_scaling = true;
    bool minimizer_progress_to_stdout = false;
    // not part of Ceres: arithmetic type of the per-point evaluation and the GPU to use
    int ea_dtype = EA_F64;
    int ea_device = 0;
  };
  struct Summary {
    TerminationType termination_type = FAILURE;
    std::string message = "ceres::Solve was not called.";
    double initial_cost = -1, final_cost = -1;
    int num_successful_steps = -1, num_unsuccessful_steps = -1;
    int num_residual_blocks = 0, num_residuals = 0;
    int num_parameter_blocks = 2, num_parameters = 7, num_effective_parameters = 6;
    double total_time_in_seconds = -1;
    ea_summary detail{};

    bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE; }
    std::string BriefReport() const {
      std::ostringstream o;
      o << "edge_alignment_amd Report: Iterations: " << (num_successful_steps + num_unsuccessful_steps)
        << ", Initial cost: " << initial_cost << ", Final cost: " << final_cost
        << ", Termination: " << TermName();
      return o.str();
    }
    std::string FullReport() const {
      char buf[2048];
      std::snprintf(buf, sizeof(buf),
                    "\nSolver Summary (edge_alignment_amd, MI355X gfx950; ceres:: facade)\n\n"
                    "Parameter blocks %26d\nParameters %32d\nEffective parameters %22d\n"
                    "Residual blocks %27d\nResidual %34d\n\n"
                    "Minimizer                        TRUST_REGION\n"
                    "Linear solver          6x6 normal equations (device)\n\n"
                    "Cost:\nInitial %35.6e\nFinal %37.6e\nChange %36.6e\n\n"
                    "Minimizer iterations %22d\nSuccessful steps %26d\nUnsuccessful steps %24d\n\n"
                    "Time (in seconds):\nTotal %37.4f\n\nTermination: %28s (%s)\n",
                    num_parameter_blocks, num_parameters, num_effective_parameters, num_residual_blocks,
                    num_residuals, initial_cost, final_cost, initial_cost - final_cost,
                    num_successful_steps + num_unsuccessful_steps, num_successful_steps, num_unsuccessful_steps,
                    total_time_in_seconds, TermName(), message.c_str());
      return std::string(buf);
    }
    const char *TermName() const {
      switch (termination_type) {
        case CONVERGENCE: return "CONVERGENCE";
        case NO_CONVERGENCE: return "NO_CONVERGENCE";
        case FAILURE: return "FAILURE";
        default: return "USER";
      }
    }
  };
};

namespace internal {
inline const char *WhyMessage(int why) {
  switch (why) {
    case EA_WHY_FUNCTION_TOL: return "Function tolerance reached.";
    case EA_WHY_GRADIENT_TOL: return "Gradient tolerance reached.";
    case EA_WHY_PARAMETER_TOL: return "Parameter tolerance reached.";
    case EA_WHY_MAX_ITERATIONS: return "Maximum number of iterations reached.";
    case EA_WHY_MIN_RADIUS: return "Minimum trust region radius reached.";
    case EA_WHY_INITIAL_EVAL_FAILED: return "Initial residual and Jacobian evaluation failed.";
    case EA_WHY_TOO_MANY_INVALID_STEPS: return "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.";
    case EA_WHY_EVAL_FAILED: return "Residual and Jacobian evaluation failed.";
    default: return "";
  }
}
}  // namespace internal

// standalone_edge_align.cpp:286  ceres::Solve(options, &problem, &summary)
inline void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary);

class ProblemAccess {  // keeps Problem's internals private to user code
 public:
  // Blocks -> residual families (same interpolator, intrinsics, functor variant and loss) -> one GPU problem per
  // family, the first one carrying the others as terms: all of them share (q, t), the way the reference adds
  // camera-1 and camera-2 blocks to one ceres::Problem (standalone_edge_align.cpp:791-803).  order[k] = indices of
  // family k's blocks in the order they were added.  Returns EA_OK, a libea_hip error code, or -1000 with *err set
  // for a problem this facade cannot host.
  static int Build(Problem *problem, int dtype, int device, std::vector<ea_problem *> *ps_out,
                   std::vector<std::vector<int>> *order, std::string *err) {
    const auto &blocks = problem->blocks_;
    const auto &b0 = blocks[0];
    struct Family { const Problem::Block *first; std::vector<double> xyz; std::vector<int> idx; };
    std::vector<Family> fams;
    for (size_t i = 0; i < blocks.size(); ++i) {
      const auto &b = blocks[i];
      if (!b.ok) { *err = "residual block is not an EAResidue-family block (this facade only hosts the edge-alignment hot path)"; return -1000; }
      if (b.q != b0.q || b.t != b0.t) { *err = "all residual blocks must share one (quaternion, translation) pair"; return -1000; }
      Family *f = nullptr;
      for (auto &cand : fams)
        if (SameFamily(*cand.first, b)) { f = &cand; break; }
      if (!f) {
        fams.push_back(Family{&b, {}, {}});
        f = &fams.back();
        f->xyz.reserve(3 * (blocks.size() - i));  // (usually the one family: no regrowth while 44 457 blocks are gathered)
        f->idx.reserve(blocks.size() - i);
      }
      f->xyz.push_back(b.X); f->xyz.push_back(b.Y); f->xyz.push_back(b.Z);
      f->idx.push_back((int)i);
    }
    if (problem->quat_param_on_ != b0.q) { *err = "the quaternion block needs QuaternionParameterization (problem.SetParameterization)"; return -1000; }
    std::vector<ea_problem *> &ps = *ps_out;
    ps.assign(fams.size(), nullptr);
    order->clear();
    int rc = EA_OK;
    for (size_t k = 0; k < fams.size() && rc == EA_OK; ++k) {
      const EABlockInfo &bi = problem->fams_[(size_t)fams[k].first->fam];
      order->push_back(fams[k].idx);
      ea_camera cam = {bi.fx, bi.fy, bi.cx, bi.cy};
      rc = ea_problem_create(&ps[k], &cam, dtype, device);
      if (rc == EA_OK) rc = ea_problem_set_points(ps[k], fams[k].xyz.data(), (int64_t)(fams[k].xyz.size() / 3), 3);
      if (rc == EA_OK) rc = ea_problem_set_dt(ps[k], bi.grid_data, bi.grid_rows, bi.grid_cols);
      if (rc == EA_OK) rc = ea_problem_set_flavour(ps[k], bi.z_guard, bi.z_eps, bi.rot_transposed);
      if (rc == EA_OK && (bi.variant & 1)) rc = ea_problem_set_distortion(ps[k], bi.dist[0], bi.dist[1], bi.dist[2], bi.dist[3], bi.dist[4]);
      if (rc == EA_OK && (bi.variant & 2)) rc = ea_problem_set_second_camera(ps[k], bi.T12, bi.T12inv);
      if (rc == EA_OK) {
        int kind = EA_LOSS_TRIVIAL; double a = 1.0;
        if (fams[k].first->loss) { kind = fams[k].first->loss->ea_kind(); a = fams[k].first->loss->ea_scale(); }
        rc = ea_problem_set_loss(ps[k], kind, a);
      }
      if (rc == EA_OK && k > 0) rc = ea_problem_add_term(ps[0], ps[k]);
    }
    return rc;
  }

  static bool Evaluate(Problem *problem, const Problem::EvaluateOptions &opt, double *cost, std::vector<double> *residuals,
                       std::vector<double> *gradient, CRSMatrix *jacobian) {
    const auto &blocks = problem->blocks_;
    if (jacobian) {
      jacobian->num_rows = (int)blocks.size();
      jacobian->num_cols = blocks.empty() ? 0 : 6;
      jacobian->rows.assign(blocks.size() + 1, 0);
      jacobian->cols.assign(blocks.size() * 6, 0);
      jacobian->values.assign(blocks.size() * 6, 0.0);
      for (size_t i = 0; i < blocks.size(); ++i) {
        jacobian->rows[i + 1] = (int)(6 * (i + 1));
        for (int a = 0; a < 6; ++a) jacobian->cols[6 * i + a] = a;
      }
    }
    if (blocks.empty()) {
      if (cost) *cost = 0.0;
      if (residuals) residuals->clear();
      if (gradient) gradient->clear();
      return true;
    }
    std::vector<ea_problem *> ps;
    std::vector<std::vector<int>> order;
    std::string err;
    int rc = Build(problem, opt.ea_dtype, opt.ea_device, &ps, &order, &err);
    const double *q = blocks[0].q, *t = blocks[0].t;
    double c = 0.0, JtJ[36], Jtr[6];
    int64_t bad = 0;
    if (rc == EA_OK) rc = ea_eval(ps[0], q, t, &c, JtJ, Jtr, &bad);  // the problem with all its terms
    if (rc == EA_OK && (residuals || jacobian)) {
      if (residuals) residuals->assign(blocks.size(), 0.0);
      for (size_t k = 0; k < ps.size() && rc == EA_OK; ++k) {
        std::vector<double> r(order[k].size()), J(jacobian ? order[k].size() * 6 : 0);
        // a term evaluated on its own: its residuals (and 1x6 rows) in the order its blocks were added
        rc = ea_eval_points(ps[k], q, t, r.data(), jacobian ? J.data() : nullptr, opt.apply_loss_function ? 1 : 0);
        for (size_t i = 0; i < r.size(); ++i) {
          if (residuals) (*residuals)[order[k][i]] = r[i];
          if (jacobian)
            for (int a = 0; a < 6; ++a) jacobian->values[6 * (size_t)order[k][i] + a] = J[6 * i + a];
        }
      }
    }
    for (auto *p : ps)
      if (p) ea_problem_destroy(p);
    if (rc != EA_OK || bad > 0) return false;  // Ceres: a failed residual block fails the evaluation
    if (cost) *cost = c;
    if (gradient) gradient->assign(Jtr, Jtr + 6);
    return true;
  }

  static void Run(const Solver::Options &options, Problem *problem, Solver::Summary *summary) {
    Solver::Summary &s = *summary;
    s = Solver::Summary();
    const auto &blocks = problem->blocks_;
    s.num_residual_blocks = s.num_residuals = (int)blocks.size();
    auto fail = [&](const std::string &m) { s.termination_type = FAILURE; s.message = m; };
    if (options.minimizer_type != TRUST_REGION) return fail("only TRUST_REGION is supported");
    if (blocks.empty()) { s.termination_type = CONVERGENCE; s.message = "No residual blocks."; s.initial_cost = s.final_cost = 0; s.num_successful_steps = s.num_unsuccessful_steps = 0; return; }
    const auto &b0 = blocks[0];
    std::vector<ea_problem *> ps;
    std::vector<std::vector<int>> order;
    std::string berr;
    int rc = Build(problem, options.ea_dtype, options.ea_device, &ps, &order, &berr);
    if (rc == -1000) return fail(berr);
    ea_options o;
    ea_default_options(&o);
    o.max_num_iterations = options.max_num_iterations;
    o.function_tolerance = options.function_tolerance;
    o.gradient_tolerance = options.gradient_tolerance;
    o.parameter_tolerance = options.parameter_tolerance;
    o.initial_trust_region_radius = options.initial_trust_region_radius;
    o.max_trust_region_radius = options.max_trust_region_radius;
    o.min_trust_region_radius = options.min_trust_region_radius;
    o.min_relative_decrease = options.min_relative_decrease;
    o.min_lm_diagonal = options.min_lm_diagonal;
    o.max_lm_diagonal = options.max_lm_diagonal;
    o.max_num_consecutive_invalid_steps = options.max_num_consecutive_invalid_steps;
    o.jacobi_scaling = options.jacobi_scaling ? 1 : 0;
    o.strategy = options.trust_region_strategy_type == DOGLEG ? EA_STRATEGY_DOGLEG : EA_STRATEGY_LM;
    o.minimizer_progress_to_stdout = options.minimizer_progress_to_stdout ? 1 : 0;
    if (rc == EA_OK) rc = ea_solve(ps[0], &o, b0.q, b0.t, &s.detail);  // q, t updated in place, like Ceres
    const std::string err = rc != EA_OK ? std::string(ea_last_error()) : std::string();
    for (size_t k = 0; k < ps.size(); ++k)
      if (ps[k]) ea_problem_destroy(ps[k]);
    if (rc != EA_OK) return fail(std::string("libea_hip: ") + err);
    s.termination_type = s.detail.termination == EA_CONVERGENCE ? CONVERGENCE : (s.detail.termination == EA_NO_CONVERGENCE ? NO_CONVERGENCE : FAILURE);
    s.message = internal::WhyMessage(s.detail.why);
    s.initial_cost = s.detail.initial_cost;
    s.final_cost = s.detail.final_cost;
    s.num_successful_steps = s.detail.num_successful_steps;
    s.num_unsuccessful_steps = s.detail.num_unsuccessful_steps;
    s.total_time_in_seconds = s.detail.total_time_ms * 1e-3;
  }

 private:
  // same functor description (by construction: same index) and an equivalent loss
  static bool SameFamily(const Problem::Block &a, const Problem::Block &b) {
    return a.fam == b.fam && (a.loss == b.loss || SameLoss(a.loss, b.loss));
  }
  static bool SameLoss(const LossFunction *a, const LossFunction *b) {
    const int ka = a ? a->ea_kind() : EA_LOSS_TRIVIAL, kb = b ? b->ea_kind() : EA_LOSS_TRIVIAL;
    if (ka != kb) return false;
    if (ka == EA_LOSS_TRIVIAL) return true;
    return a->ea_scale() == b->ea_scale();
  }
};

inline bool Problem::Evaluate(const EvaluateOptions &opt, double *cost, std::vector<double> *residuals,
                              std::vector<double> *gradient, CRSMatrix *jacobian) {
  return ProblemAccess::Evaluate(this, opt, cost, residuals, gradient, jacobian);
}

inline void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary) {
  ProblemAccess::Run(options, problem, summary);
}

}  // namespace ceres

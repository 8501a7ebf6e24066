// ceres/loss_function.h — loss carriers for the ceres:: facade (see ceres.h).  The rho functions
// themselves are evaluated per point inside the HIP kernels (IRLS weight); these classes only
// name the loss and its scale.  ref: standalone_edge_align.cpp:272 `new CauchyLoss(1.)`,
// :2604 `new TrivialLoss()`, src/SolveEA.cpp:144 `new ceres::HuberLoss(0.1)`.
// Evaluate(s, rho) is Ceres' interface (rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s), s = squared residual norm):
// provided for host-side probes with Ceres' published formulas; the solver never calls it.
#pragma once
#include <cmath>

#include "../../../include/ea_hip.h"

namespace ceres {

enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };

class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double s, double rho[3]) const = 0;
  virtual int ea_kind() const = 0;
  virtual double ea_scale() const { return 1.0; }
};

class TrivialLoss : public LossFunction {
 public:
  void Evaluate(double s, double rho[3]) const override { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  int ea_kind() const override { return EA_LOSS_TRIVIAL; }
};

class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : a_(a), b_(a * a), c_(1.0 / (a * a)) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = inv > 0.0 ? inv : 0.0;   // (ceres: max(min, inv))
    rho[2] = -c_ * (inv * inv);
  }
  int ea_kind() const override { return EA_LOSS_CAUCHY; }
  double ea_scale() const override { return a_; }

 private:
  double a_, b_, c_;
};

class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = a_ / r > 0.0 ? a_ / r : 0.0;
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }
  int ea_kind() const override { return EA_LOSS_HUBER; }
  double ea_scale() const override { return a_; }

 private:
  double a_, b_;
};

// A loss whose wrapped function can be replaced between solves (the reference pulls the name into scope:
// include/EAResidue.h:31, include/SolveEA.h:29).  NULL = trivial.  The facade reads kind and scale when Solve runs.
class LossFunctionWrapper : public LossFunction {
 public:
  LossFunctionWrapper(LossFunction *rho, Ownership ownership) : rho_(rho), ownership_(ownership) {}
  LossFunctionWrapper(const LossFunctionWrapper &) = delete;
  LossFunctionWrapper &operator=(const LossFunctionWrapper &) = delete;
  ~LossFunctionWrapper() override {
    if (ownership_ == TAKE_OWNERSHIP) delete rho_;
  }
  void Reset(LossFunction *rho, Ownership ownership) {
    if (ownership_ == TAKE_OWNERSHIP) delete rho_;
    rho_ = rho;
    ownership_ = ownership;
  }
  void Evaluate(double s, double rho[3]) const override {
    if (rho_) rho_->Evaluate(s, rho);
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
  int ea_kind() const override { return rho_ ? rho_->ea_kind() : (int)EA_LOSS_TRIVIAL; }
  double ea_scale() const override { return rho_ ? rho_->ea_scale() : 1.0; }

 private:
  LossFunction *rho_;
  Ownership ownership_;
};

}  // namespace ceres

// ceres/loss_function.h — loss carriers for the ceres:: facade (see ceres.h).  The rho functions
// themselves are evaluated per point inside the HIP kernels (IRLS weight); these classes only
// name the loss and its scale.  ref: standalone_edge_align.cpp:272 `new CauchyLoss(1.)`,
// :2604 `new TrivialLoss()`, src/SolveEA.cpp:144 `new ceres::HuberLoss(0.1)`.
#pragma once
#include "../../../include/ea_hip.h"

namespace ceres {

class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual int ea_kind() const = 0;
  virtual double ea_scale() const { return 1.0; }
};

class TrivialLoss : public LossFunction {
 public:
  int ea_kind() const override { return EA_LOSS_TRIVIAL; }
};

class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : a_(a) {}
  int ea_kind() const override { return EA_LOSS_CAUCHY; }
  double ea_scale() const override { return a_; }

 private:
  double a_;
};

class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a) {}
  int ea_kind() const override { return EA_LOSS_HUBER; }
  double ea_scale() const override { return a_; }

 private:
  double a_;
};

}  // namespace ceres

// ros/SolveEA.h — drop-in for include/SolveEA.h:36-71 + src/SolveEA.cpp of kuwt/edge_alignment.
//
// Public surface kept: SolveEA(), setRefFrame(rgb, depth), setNowFrame(rgb, depth), setAsCERESProblem(),
// _verify3dPts(), _sampleCERESProblem()  (src/ea.cpp:184-191 calls them in that order).
//
// Frames stay on the device.  setRefFrame / setNowFrame hand the frame to the GPU producers
// (ea_problem_set_ref_frame_ros / ea_problem_set_now_frame_ros: Canny(150, 100, 3, true) on the 3-channel image, every
// edge pixel back-projected with Z == 0 -> 1; 255 - edges -> DIST_MASK_PRECISE -> [0, 255]) of ONE ea_problem this
// object owns; the edge points and the distance transform are written where the solve reads them, and
// setAsCERESProblem() — upstream's residual blocks, NULL loss, quaternion parameterisation and DOGLEG solve with 25
// iterations (src/SolveEA.cpp:124-216) — is one ea_solve on that problem.  Nothing is copied back to the host between
// the three calls (round 1 read points and DT back and uploaded them again through the ceres:: facade).
//
// The frame arguments are templates over "anything shaped like a cv::Mat" (.data, .rows, .cols, .isContinuous()):
// with OpenCV present `setRefFrame(const cv::Mat&, const cv::Mat&)` binds to them unchanged; this header needs no OpenCV.
// rgb: bgr8 (CV_8UC3), depth: float32 metres (CV_32F), NaN already zeroed by the caller as upstream's callback does.
// With halvings() > 0 the frames are taken at the resolution the ROS callbacks RECEIVE them and the node's
// cv::resize(..., 0.5, 0.5) (src/ea.cpp:38, :56-62: NaN -> 0 first on depth) runs on the device too.
//
// setRefPoints / setNowDistanceTransform keep the host-data path (list_edge_ref / now_dist_transform_eig filled by the
// caller): those go through the ceres:: facade as upstream's setAsCERESProblem body does.
// Upstream never returns the pose: getPose() / summary() are additions.
#pragma once
#include <cstdio>
#include <vector>

#include "EAResidue.h"

class SolveEA {
 public:
  struct Intrinsics {
    double m[9];
    double operator()(int i, int j) const { return m[3 * i + j]; }
  };

  // src/SolveEA.cpp:5-24 — TUM intrinsics at half resolution
  SolveEA() {
    fx = .5 * 525.0; fy = .5 * 525.0; cx = .5 * 319.5; cy = .5 * 239.5;
    for (double &v : K.m) v = 0;
    K.m[0] = fx; K.m[4] = fy; K.m[2] = cx; K.m[5] = cy; K.m[8] = 1.0;
    q_[0] = 1; q_[1] = q_[2] = q_[3] = 0; t_[0] = t_[1] = t_[2] = 0;
  }
  ~SolveEA() { if (dev_) ea_problem_destroy(dev_); }
  SolveEA(const SolveEA &) = delete;
  SolveEA &operator=(const SolveEA &) = delete;

  // src/SolveEA.cpp:29-82 / :86-119 with cv::Mat-shaped arguments
  template <typename Mat>
  void setRefFrame(const Mat &rgb, const Mat &depth) {
    if (!rgb.isContinuous() || !depth.isContinuous()) { std::fprintf(stderr, "SolveEA::setRefFrame: continuous frames expected\n"); return; }
    if (!setRefFrame(reinterpret_cast<const unsigned char *>(rgb.data), reinterpret_cast<const float *>(depth.data), rgb.rows, rgb.cols))
      std::fprintf(stderr, "SolveEA::setRefFrame: %s\n", ea_last_error());
  }
  template <typename Mat>
  void setNowFrame(const Mat &rgb, const Mat &depth) {
    if (!rgb.isContinuous()) { std::fprintf(stderr, "SolveEA::setNowFrame: continuous frame expected\n"); return; }
    if (!setNowFrame(reinterpret_cast<const unsigned char *>(rgb.data), reinterpret_cast<const float *>(depth.data), rgb.rows, rgb.cols))
      std::fprintf(stderr, "SolveEA::setNowFrame: %s\n", ea_last_error());
  }

  // the same two calls on raw buffers.  bgr: rows x cols x 3 bytes (bgr8, src/ea.cpp:34), depth: rows x cols float32
  // metres; rows x cols is the resolution of the buffers (full resolution when halvings() > 0).
  // Return false when the library reports an error (ea_last_error()), e.g. a frame without edges.
  bool setRefFrame(const unsigned char *bgr, const float *depth, int rows, int cols) {
    if (!device_problem()) return false;
    host_points_ = false;
    have_ref_ = ea_problem_set_ref_frame_ros_scaled(dev_, bgr, depth, rows, cols, halvings_, 150, 100) == EA_OK;
    return have_ref_;
  }
  bool setNowFrame(const unsigned char *bgr, const float * /*depth*/, int rows, int cols) {
    if (!device_problem()) return false;
    host_dt_ = false;
    have_now_ = ea_problem_set_now_frame_ros_scaled(dev_, bgr, rows, cols, halvings_, 150, 100) == EA_OK;
    return have_now_;
  }
  // frames arrive at 2^n times the working resolution: reduce them on the device first (src/ea.cpp:38, :62: n = 1)
  void setHalvings(int n) { halvings_ = n < 0 ? 0 : n; }
  int halvings() const { return halvings_; }

  // host-data path — list_edge_ref: 3 x N, column-major (src/SolveEA.cpp:55,73-75)
  void setRefPoints(const double *xyz_3xN, int N) {
    list_edge_ref.assign(xyz_3xN, xyz_3xN + 3 * (size_t)N);
    host_points_ = true;
  }
  // now_dist_transform_eig: rows x cols, column-major like Eigen::MatrixXd (src/SolveEA.cpp:110)
  void setNowDistanceTransform(const double *colmajor, int rows, int cols) {
    dt_rows = rows; dt_cols = cols;
    now_dist_transform_eig.assign(colmajor, colmajor + (size_t)rows * cols);
    host_dt_ = true;
  }

  // src/SolveEA.cpp:124-216
  void setAsCERESProblem() {
    if (!host_points_ && !host_dt_ && have_ref_ && have_now_) { solve_on_device(); return; }
    if (!host_points_ || !host_dt_) {
      // one input lives on the device, the other came from the host: bring the device one over (rare, explicit)
      if (!host_points_ && have_ref_) fetch_points();
      if (!host_dt_ && have_now_) fetch_dt();
    }
    double q_cap[4] = {1, 0, 0, 0};
    double t_cap[3] = {0, 0, 0};
    ceres::Problem problem;
    // column-major rows x cols storage viewed row-major as (cols x rows): value(r=u, c=v) = DT(v,u)
    // (the single-channel view of standalone_edge_align.cpp:258; upstream :152 is broken, App. D 2)
    ceres::Grid2D<double, 1> grid(now_dist_transform_eig.data(), 0, dt_cols, 0, dt_rows);
    ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interpolated_cost_function(grid);
    const int N = (int)(list_edge_ref.size() / 3);
    for (int ir = 0; ir < N; ir++)
      problem.AddResidualBlock(new ceres::AutoDiffCostFunction<EAResidue, 1, 4, 3>(new EAResidue(
                                   list_edge_ref[3 * ir], list_edge_ref[3 * ir + 1], list_edge_ref[3 * ir + 2],
                                   interpolated_cost_function, K)),
                               NULL, q_cap, t_cap);
    problem.SetParameterization(q_cap, new ceres::QuaternionParameterization);
    ceres::Solver::Options options;
    options.max_num_iterations = 25;
    options.linear_solver_type = ceres::DENSE_QR;
    options.minimizer_progress_to_stdout = verbose;
    options.minimizer_type = ceres::TRUST_REGION;
    options.trust_region_strategy_type = ceres::DOGLEG;
    ceres::Solve(options, &problem, &summary_);
    if (verbose) std::printf("%s\n", summary_.FullReport().c_str());
    for (int i = 0; i < 4; i++) q_[i] = q_cap[i];
    for (int i = 0; i < 3; i++) t_[i] = t_cap[i];
  }

  // debug visualisation upstream (imshow); nothing to show without a GUI
  void _verify3dPts() {}
  // upstream: random 34x27 least squares proving Ceres links; here: proves the GPU library links
  void _sampleCERESProblem() {
    int n = 0;
    ea_device_count(&n);
    std::printf("libea_hip: %s, %d gfx950 device(s)\n", ea_version(), n);
  }

  void getPose(double q[4], double t[3]) const {
    for (int i = 0; i < 4; i++) q[i] = q_[i];
    for (int i = 0; i < 3; i++) t[i] = t_[i];
  }
  const ceres::Solver::Summary &summary() const { return summary_; }
  int numRefPoints() const {
    return (!host_points_ && dev_ && have_ref_) ? (int)ea_problem_num_points(dev_) : (int)(list_edge_ref.size() / 3);
  }
  bool verbose = false;

 private:
  bool device_problem() {
    if (dev_) return true;
    const ea_camera cam = {fx, fy, cx, cy};
    if (ea_problem_create(&dev_, &cam, EA_F64, 0) != EA_OK) { dev_ = nullptr; return false; }
    // this flavour's functor: R applied transposed, divisor z + 0.001, no guard (include/EAResidue.h:90-105 upstream);
    // loss NULL (src/SolveEA.cpp:171)
    return ea_problem_set_flavour(dev_, 0.0, 0.001, 1) == EA_OK && ea_problem_set_loss(dev_, EA_LOSS_TRIVIAL, 1.0) == EA_OK;
  }
  // setAsCERESProblem's set-up block as options of one device solve (src/SolveEA.cpp:130-131, :184-198)
  void solve_on_device() {
    double q_cap[4] = {1, 0, 0, 0};
    double t_cap[3] = {0, 0, 0};
    ea_options o;
    ea_default_options(&o);
    o.max_num_iterations = 25;
    o.strategy = EA_STRATEGY_DOGLEG;
    o.minimizer_progress_to_stdout = verbose ? 1 : 0;
    ceres::Solver::Summary s;
    s.num_residual_blocks = s.num_residuals = (int)ea_problem_num_points(dev_);
    if (ea_solve(dev_, &o, q_cap, t_cap, &s.detail) != EA_OK) {
      s.termination_type = ceres::FAILURE;
      s.message = std::string("libea_hip: ") + ea_last_error();
    } else {
      s.termination_type = s.detail.termination == EA_CONVERGENCE ? ceres::CONVERGENCE
                           : (s.detail.termination == EA_NO_CONVERGENCE ? ceres::NO_CONVERGENCE : ceres::FAILURE);
      s.message = ceres::internal::WhyMessage(s.detail.why);
      s.initial_cost = s.detail.initial_cost;
      s.final_cost = s.detail.final_cost;
      s.num_successful_steps = s.detail.num_successful_steps;
      s.num_unsuccessful_steps = s.detail.num_unsuccessful_steps;
      s.total_time_in_seconds = s.detail.total_time_ms * 1e-3;
    }
    summary_ = s;
    if (verbose) std::printf("%s\n", summary_.FullReport().c_str());
    for (int i = 0; i < 4; i++) q_[i] = q_cap[i];
    for (int i = 0; i < 3; i++) t_[i] = t_cap[i];
  }
  void fetch_points() {
    const long long n = ea_problem_num_points(dev_);
    std::vector<double> pts((size_t)3 * (size_t)n);
    if (ea_problem_get_points(dev_, pts.data(), n) == EA_OK) setRefPoints(pts.data(), (int)n);
  }
  void fetch_dt() {
    int h = 0, w = 0;
    if (ea_problem_get_dt(dev_, nullptr, &h, &w) != EA_OK) return;
    std::vector<double> img((size_t)h * w), colmajor((size_t)h * w);
    if (ea_problem_get_dt(dev_, img.data(), nullptr, nullptr) != EA_OK) return;
    for (int c = 0; c < w; ++c)
      for (int r = 0; r < h; ++r) colmajor[(size_t)c * h + r] = img[(size_t)r * w + c];
    setNowDistanceTransform(colmajor.data(), h, w);
  }

  Intrinsics K;
  double fx, fy, cx, cy;
  ea_problem *dev_ = nullptr;
  bool have_ref_ = false, have_now_ = false, host_points_ = false, host_dt_ = false;
  int halvings_ = 0;
  std::vector<double> list_edge_ref;
  std::vector<double> now_dist_transform_eig;
  int dt_rows = 0, dt_cols = 0;
  double q_[4], t_[3];
  ceres::Solver::Summary summary_;
};

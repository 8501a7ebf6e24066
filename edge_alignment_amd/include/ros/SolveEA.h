// ros/SolveEA.h — drop-in for include/SolveEA.h:36-71 + src/SolveEA.cpp of kuwt/edge_alignment.
//
// Public surface kept: SolveEA(), setRefFrame(rgb, depth), setNowFrame(rgb, depth),
// setAsCERESProblem(), _verify3dPts(), _sampleCERESProblem()  (src/ea.cpp:184-191 calls them in
// that order).  The cv::Mat overloads exist only when OpenCV headers are present (they are not in
// this image) and keep upstream's OpenCV calls.  The raw-buffer overloads of setRefFrame / setNowFrame
// (bgr8 + float32 depth, what the cv::Mat arguments hold) run the same pre-processing on the GPU
// (ea_problem_set_ref_frame_ros / ea_problem_set_now_frame_ros: Canny(150, 100, 3, true) on the 3-channel
// image, DIST_MASK_PRECISE, normalisation to [0, 255], Z == 0 -> 1) and feed the same members.
// setAsCERESProblem() — the residual blocks, the loss, the parameterisation and the DOGLEG solve
// (src/SolveEA.cpp:124-216) — runs on the GPU through the ceres:: facade.
// Upstream never returns the pose (App. D 4): getPose()/summary() are additions.
#pragma once
#include <cstdio>
#include <vector>

#include "EAResidue.h"

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && __has_include(<opencv2/imgproc/imgproc.hpp>)
#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#define EA_HAVE_OPENCV 1
#endif
#endif

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

#ifdef EA_HAVE_OPENCV
  // src/SolveEA.cpp:29-82
  void setRefFrame(const cv::Mat &rgb, const cv::Mat &depth) {
    cv::Mat edge;
    cv::Canny(rgb, edge, 150, 100, 3, true);
    std::vector<double> pts;
    for (int yy = 0; yy < rgb.rows; yy++)
      for (int xx = 0; xx < rgb.cols; xx++)
        if (edge.at<uchar>(yy, xx) > 0) {
          double Z = depth.at<float>(yy, xx);
          Z = (Z == 0) ? 1.0 : Z;
          pts.push_back(Z * (xx - cx) / fx); pts.push_back(Z * (yy - cy) / fy); pts.push_back(Z);
        }
    setRefPoints(pts.data(), (int)(pts.size() / 3));
  }
  // src/SolveEA.cpp:86-119
  void setNowFrame(const cv::Mat &rgb, const cv::Mat & /*depth*/) {
    cv::Mat edge, dist;
    cv::Canny(rgb, edge, 150, 100, 3, true);
    edge = 255 - edge;
    cv::distanceTransform(edge, dist, cv::DIST_L2, cv::DIST_MASK_PRECISE);
    cv::normalize(dist, dist, 0.0, 255.0, cv::NORM_MINMAX);
    std::vector<double> colmajor((size_t)dist.rows * dist.cols);
    for (int c = 0; c < dist.cols; ++c)
      for (int r = 0; r < dist.rows; ++r) colmajor[(size_t)c * dist.rows + r] = dist.at<float>(r, c);
    setNowDistanceTransform(colmajor.data(), dist.rows, dist.cols);
  }
#endif

  // OpenCV-free forms of the same two calls (src/SolveEA.cpp:29-82, :86-119), pre-processing on the GPU.
  // bgr: rows x cols x 3 bytes (bgr8, src/ea.cpp:34), depth: rows x cols float32 metres (NaN already set to 0, :56-58).
  // Return false (and leave the members untouched) when the library reports an error, e.g. a frame without edges.
  bool setRefFrame(const unsigned char *bgr, const float *depth, int rows, int cols) {
    ea_problem *p = nullptr;
    const ea_camera cam = {fx, fy, cx, cy};
    if (ea_problem_create(&p, &cam, EA_F64, 0) != EA_OK) return false;
    bool ok = ea_problem_set_ref_frame_ros(p, bgr, depth, rows, cols, 150, 100) == EA_OK;
    if (ok) {
      const long long n = ea_problem_num_points(p);
      std::vector<double> pts((size_t)3 * (size_t)n);
      ok = ea_problem_get_points(p, pts.data(), n) == EA_OK;
      if (ok) setRefPoints(pts.data(), (int)n);
    }
    ea_problem_destroy(p);
    return ok;
  }
  bool setNowFrame(const unsigned char *bgr, const float * /*depth*/, int rows, int cols) {
    ea_problem *p = nullptr;
    const ea_camera cam = {fx, fy, cx, cy};
    if (ea_problem_create(&p, &cam, EA_F64, 0) != EA_OK) return false;
    bool ok = ea_problem_set_now_frame_ros(p, bgr, rows, cols, 150, 100) == EA_OK;
    if (ok) {
      std::vector<double> img((size_t)rows * cols), colmajor((size_t)rows * cols);
      int h = 0, w = 0;
      ok = ea_problem_get_dt(p, img.data(), &h, &w) == EA_OK && h == rows && w == cols;
      if (ok) {
        for (int c = 0; c < cols; ++c)
          for (int r = 0; r < rows; ++r) colmajor[(size_t)c * rows + r] = img[(size_t)r * cols + c];
        setNowDistanceTransform(colmajor.data(), rows, cols);
      }
    }
    ea_problem_destroy(p);
    return ok;
  }

  // list_edge_ref: 3 x N, column-major (src/SolveEA.cpp:55,73-75)
  void setRefPoints(const double *xyz_3xN, int N) { list_edge_ref.assign(xyz_3xN, xyz_3xN + 3 * (size_t)N); }
  // now_dist_transform_eig: rows x cols, column-major like Eigen::MatrixXd (src/SolveEA.cpp:110)
  void setNowDistanceTransform(const double *colmajor, int rows, int cols) {
    dt_rows = rows; dt_cols = cols;
    now_dist_transform_eig.assign(colmajor, colmajor + (size_t)rows * cols);
  }

  // src/SolveEA.cpp:124-216
  void setAsCERESProblem() {
    double q_cap[4] = {1, 0, 0, 0};
    double t_cap[3] = {0, 0, 0};
    ceres::Problem problem;
    // column-major rows x cols storage viewed row-major as (cols x rows): value(r=u, c=v) = DT(v,u)
    // (the single-channel view of standalone_edge_align.cpp:258; upstream :152 is broken, App. D 2)
    ceres::Grid2D<double, 1> grid(now_dist_transform_eig.data(), 0, dt_cols, 0, dt_rows);
    ceres::BiCubicInterpolator<ceres::Grid2D<double, 1>> interpolated_cost_function(grid);
    const int N = (int)(list_edge_ref.size() / 3);
    for (int ir = 0; ir < N; ir++) {
      const double curX = list_edge_ref[3 * ir], curY = list_edge_ref[3 * ir + 1], curZ = list_edge_ref[3 * ir + 2];
      problem.AddResidualBlock(new ceres::AutoDiffCostFunction<EAResidue, 1, 4, 3>(
                                   new EAResidue(curX, curY, curZ, interpolated_cost_function, K)),
                               NULL, q_cap, t_cap);
    }
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
  int numRefPoints() const { return (int)(list_edge_ref.size() / 3); }
  bool verbose = false;

 private:
  Intrinsics K;
  double fx, fy, cx, cy;
  std::vector<double> list_edge_ref;
  std::vector<double> now_dist_transform_eig;
  int dt_rows = 0, dt_cols = 0;
  double q_[4], t_[3];
  ceres::Solver::Summary summary_;
};

// examples/ros_frames_demo.cpp — the call sequence of src/ea.cpp:184-191 (setRefFrame, setNowFrame, _verify3dPts,
// setAsCERESProblem) on raw frames: bgr8 + float32 depth buffers, what the ROS callbacks hold after
// cv_bridge / resize (src/ea.cpp:30-64).  Pre-processing and solve both run on the GPU.
// input file: int32 rows, cols | ref bgr (rows*cols*3 bytes) | ref depth (rows*cols float32) | now bgr
// usage: ros_frames_demo frames.bin [halvings]   (halvings > 0: the file holds full-resolution frames)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ros/SolveEA.h"

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s frames.bin\n", argv[0]); return 2; }
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t rows, cols;
  if (std::fread(&rows, 4, 1, f) != 1 || std::fread(&cols, 4, 1, f) != 1) return 2;
  const size_t np = (size_t)rows * cols;
  std::vector<unsigned char> ref_im(np * 3), now_im(np * 3);
  std::vector<float> ref_depth(np);
  if (std::fread(ref_im.data(), 1, np * 3, f) != np * 3) return 2;
  if (std::fread(ref_depth.data(), 4, np, f) != np) return 2;
  if (std::fread(now_im.data(), 1, np * 3, f) != np * 3) return 2;
  std::fclose(f);
  // a stand-in with cv::Mat's shape (data, rows, cols, isContinuous()): the calls below are src/ea.cpp:184-191 verbatim
  struct Mat {
    unsigned char *data; int rows, cols;
    bool isContinuous() const { return true; }
  };
  Mat ref_im_m = {ref_im.data(), rows, cols}, now_im_m = {now_im.data(), rows, cols};
  Mat ref_depth_m = {reinterpret_cast<unsigned char *>(ref_depth.data()), rows, cols};
  SolveEA *ea = new SolveEA();
  if (argc > 2) ea->setHalvings(std::atoi(argv[2]));   // frames at 2^n x the working resolution: resized on the device
  ea->setRefFrame(ref_im_m, ref_depth_m);
  ea->setNowFrame(now_im_m, ref_depth_m);
  if (ea->numRefPoints() == 0) { std::fprintf(stderr, "no reference points: %s\n", ea_last_error()); return 1; }
  ea->_verify3dPts();
  ea->setAsCERESProblem();
  double q[4], t[3];
  ea->getPose(q, t);
  std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d\n", q[0], q[1], q[2], q[3], t[0], t[1], t[2],
              (int)ea->summary().termination_type, ea->numRefPoints());
  delete ea;
  return 0;
}

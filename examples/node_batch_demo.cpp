// node_batch_demo.cpp — the multi-GPU mode of the path in the reference's language: batches of independent frame-pair
// problems, one batch per GPU, solved by host threads of ONE process, and a single RCCL gather of the 6-DoF poses.
//
//   node_batch_demo problem.bin pairs_per_gpu [ndev]
//
// problem.bin: the layout the other examples read (int32 N, H, W; double fx, fy, cx, cy; a_X as 4 x N column-major
// doubles; the Grid2D view W x H row-major doubles -- standalone_edge_align.cpp:169-206, :258).  Every GPU gets
// `pairs_per_gpu` problems on that data; problem g (global index, rank-major) starts from a rotation of 0.02 deg * g about
// the optical axis, so that the gathered poses tell the problems apart.  The unit of sharding is what the reference
// hands to one ceres::Solve (standalone_edge_align.cpp:286); nothing is exchanged during the solves.
//
// stdout: one line per problem of the gathered result as rank 0 holds it -- `g q0 q1 q2 q3 t0 t1 t2 termination` with 17
// significant digits -- after a header line `ndev pairs_per_gpu solve_ms_max gather_ms_max same_on_every_rank`.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../include/ea_hip.h"

struct Shared {
  int N = 0, H = 0, W = 0, pairs = 0, ndev = 0;
  ea_camera cam{};
  std::vector<double> aX, grid;
};

struct RankResult {
  int rc = EA_OK;
  std::string error;
  double solve_ms = 0.0, gather_ms = 0.0;
  std::vector<double> all_q, all_t;
  std::vector<int> all_status;
};

static void run_rank(const Shared &sh, ea_comm *comm, int device, RankResult *out) {
  const int rank = ea_comm_rank(comm), m = sh.pairs;
  std::vector<ea_problem *> probs((size_t)m, nullptr);
  ea_batch *batch = nullptr;
  int rc = EA_OK;
  for (int i = 0; i < m && rc == EA_OK; ++i) {
    rc = ea_problem_create(&probs[(size_t)i], &sh.cam, EA_F64, device);
    if (rc == EA_OK) rc = ea_problem_set_points(probs[(size_t)i], sh.aX.data(), sh.N, 4);  // the 4 x N a_X of get_aX, stride 4
    if (rc == EA_OK) rc = ea_problem_set_dt(probs[(size_t)i], sh.grid.data(), sh.W, sh.H);
    if (rc == EA_OK) rc = ea_problem_set_loss(probs[(size_t)i], EA_LOSS_CAUCHY, 1.0);
  }
  if (rc == EA_OK) rc = ea_batch_create(&batch, probs.data(), m);
  std::vector<double> q((size_t)m * 4), t((size_t)m * 3, 0.0);
  std::vector<ea_summary> sums((size_t)m);
  for (int i = 0; i < m; ++i) {
    const double half = 0.5 * (0.02 * (rank * m + i)) * M_PI / 180.0;
    q[4 * (size_t)i] = std::cos(half); q[4 * (size_t)i + 1] = 0.0; q[4 * (size_t)i + 2] = 0.0; q[4 * (size_t)i + 3] = std::sin(half);
  }
  ea_options opt;
  ea_default_options(&opt);
  const auto t0 = std::chrono::steady_clock::now();
  if (rc == EA_OK) rc = ea_batch_solve(batch, &opt, q.data(), t.data(), sums.data());
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<int> status((size_t)m);
  for (int i = 0; i < m; ++i) status[(size_t)i] = sums[(size_t)i].termination;
  const int n = ea_comm_size(comm) * m;
  out->all_q.assign((size_t)n * 4, 0.0); out->all_t.assign((size_t)n * 3, 0.0); out->all_status.assign((size_t)n, -1);
  // THE collective: one ncclAllGather of m x 8 doubles on the batch's stream.  Every rank must reach it, also one whose
  // solve failed (it contributes what it has), or the others would wait for ever.
  const int grc = ea_comm_gather_poses(comm, rc == EA_OK ? batch : nullptr, q.data(), t.data(), status.data(), m, out->all_q.data(),
                                       out->all_t.data(), out->all_status.data());
  const auto t2 = std::chrono::steady_clock::now();
  if (rc == EA_OK) rc = grc;
  if (rc != EA_OK) out->error = ea_last_error();
  out->rc = rc;
  out->solve_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  out->gather_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
  if (batch) ea_batch_destroy(batch);
  for (ea_problem *p : probs) ea_problem_destroy(p);
}

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s problem.bin pairs_per_gpu [ndev]\n", argv[0]); return 2; }
  Shared sh;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int hdr[3];
  double K[4];
  if (std::fread(hdr, sizeof(int), 3, f) != 3 || std::fread(K, sizeof(double), 4, f) != 4) return 2;
  sh.N = hdr[0]; sh.H = hdr[1]; sh.W = hdr[2];
  sh.cam.fx = K[0]; sh.cam.fy = K[1]; sh.cam.cx = K[2]; sh.cam.cy = K[3];
  sh.aX.resize((size_t)sh.N * 4); sh.grid.resize((size_t)sh.H * sh.W);
  if (std::fread(sh.aX.data(), sizeof(double), sh.aX.size(), f) != sh.aX.size() ||
      std::fread(sh.grid.data(), sizeof(double), sh.grid.size(), f) != sh.grid.size()) return 2;
  std::fclose(f);
  sh.pairs = std::atoi(argv[2]);
  int visible = 0;
  if (ea_device_count(&visible) != EA_OK || visible < 1) { std::fprintf(stderr, "no gfx950 device: %s\n", ea_last_error()); return 1; }
  sh.ndev = argc > 3 ? std::atoi(argv[3]) : visible;
  if (sh.pairs < 1 || sh.ndev < 1 || sh.ndev > visible) { std::fprintf(stderr, "bad pairs_per_gpu / ndev\n"); return 2; }

  std::vector<ea_comm *> comms((size_t)sh.ndev, nullptr);
  if (ea_comm_create_all(comms.data(), nullptr, sh.ndev) != EA_OK) { std::fprintf(stderr, "ea_comm_create_all: %s\n", ea_last_error()); return 1; }
  std::vector<RankResult> res((size_t)sh.ndev);
  std::vector<std::thread> th;
  for (int d = 0; d < sh.ndev; ++d) th.emplace_back(run_rank, std::cref(sh), comms[(size_t)d], d, &res[(size_t)d]);
  for (std::thread &x : th) x.join();
  for (ea_comm *c : comms) ea_comm_destroy(c);

  double solve_ms = 0.0, gather_ms = 0.0;
  int same = 1;
  for (int d = 0; d < sh.ndev; ++d) {
    if (res[(size_t)d].rc != EA_OK) { std::fprintf(stderr, "rank %d: libea_hip error %d: %s\n", d, res[(size_t)d].rc, res[(size_t)d].error.c_str()); return 1; }
    solve_ms = std::fmax(solve_ms, res[(size_t)d].solve_ms);
    gather_ms = std::fmax(gather_ms, res[(size_t)d].gather_ms);
    same = same && res[(size_t)d].all_q == res[0].all_q && res[(size_t)d].all_t == res[0].all_t && res[(size_t)d].all_status == res[0].all_status;
  }
  std::printf("%d %d %.6f %.6f %d\n", sh.ndev, sh.pairs, solve_ms, gather_ms, same);
  const RankResult &r0 = res[0];
  for (int g = 0; g < sh.ndev * sh.pairs; ++g)
    std::printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n", g, r0.all_q[4 * (size_t)g], r0.all_q[4 * (size_t)g + 1], r0.all_q[4 * (size_t)g + 2],
                r0.all_q[4 * (size_t)g + 3], r0.all_t[3 * (size_t)g], r0.all_t[3 * (size_t)g + 1], r0.all_t[3 * (size_t)g + 2], r0.all_status[(size_t)g]);
  return 0;
}

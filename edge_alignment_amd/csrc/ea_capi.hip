// ea_capi.hip — host driver behind include/ea_hip.h: HBM residency, tile lists, launches,
// the device-resident trust-region loop, and the measurement hooks.  No CPU compute fallback:
// without a gfx950 device every compute entry point returns EA_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ea_hip.h"
#include "ea_hip_dev.h"
#include "ea_lm.h"
#include "ea_spin.h"
#include "ea_types.h"

namespace ea {
hipError_t launch_eval_fused(int dtype, int ppt, int nt, int variant, const ProblemDesc *probs, int nterms, int chunk,
                             int max_chunks, int xcd_remap, const PoseState *poses, double *partials,
                             int lds_bytes, int wide, int terms_are_groups, int buffer_loads, int img32, const void *x0, const void *y0,
                             const void *z0, int n0, hipStream_t stream);
hipError_t launch_eval_poses(int dtype, int ppt, int nt, int variant, const ProblemDesc *probs, int nterms, int chunk,
                             int max_chunks, int xcd_remap, const PoseState *poses, double *partials,
                             int lds_bytes, int wide, int terms_are_groups, int buffer_loads, int img32, const void *x0, const void *y0,
                             const void *z0, int n0, hipStream_t stream);
hipError_t launch_pixel_cost(int dtype, const ProblemDesc *probs, int problem, int n, const PoseState *poses, void *partials,
                             hipStream_t stream);
hipError_t launch_eval_rows(int dtype, int variant, int buffer_loads, int img32, int layout, int staged, const ProblemDesc *probs, int nterms,
                            long long max_n, const PoseState *poses, int corrected, int nontemporal, long long total_rows,
                            void *r_out, void *J_out, unsigned int *n_invalid, hipStream_t stream);
hipError_t launch_eval_points(int dtype, const ProblemDesc *probs, int problem, int n, const PoseState *poses,
                              double *r_out, double *J_out, int corrected, hipStream_t stream);
hipError_t launch_reduce(const GroupDesc *groups, int count, const double *partials, EvalOut *out,
                         hipStream_t stream);
hipError_t launch_reduce_done(const GroupDesc *groups, int count, const double *partials, EvalOut *out, unsigned int *counter,
                              int *host_flag, int seq, hipStream_t stream);
hipError_t launch_eval_fold(int dtype, int ppt, int nt, const ProblemDesc *probs, int nterms, int chunk, int max_chunks,
                            int xcd_remap, const PoseState *poses, double *partials, int buffer_loads, int img32, const void *x0,
                            const void *y0, const void *z0, int n0, const GroupDesc *groups, const double *prev_rows,
                            EvalOut *prev_out, hipStream_t stream);
hipError_t launch_reduce_nt(int nt, const GroupDesc *groups, int count, const double *partials, EvalOut *out,
                            hipStream_t stream);
hipError_t launch_lm_step(const GroupDesc *groups, int count, const double *partials, PoseState *poses,
                          LMState *states, LMCold *cold, LMTrace *traces, const LMOptions &opt, int *running_flags,
                          LMState *host_states, LMTrace *host_traces, const GroupDesc &first, int post_done, hipStream_t stream);
hipError_t launch_lm_iter(int dtype, int ppt, const ProblemDesc *probs, int count, int chunk, int max_chunks, int xcd_remap,
                          PoseState *poses, const double *rows_in, double *rows_out, int buffer_loads, int img32,
                          const void *x0, const void *y0, const void *z0, int n0, const GroupDesc *groups,
                          const LMState *st_in, LMState *st_out, const LMCold *cold_in, LMCold *cold_out, LMTrace *traces,
                          const LMOptions &opt, int *progress, LMState *host_states, LMTrace *host_traces,
                          const GroupDesc &first, int post_done, hipStream_t stream);
hipError_t launch_pad_image(int dtype, const void *src, int H, int W, void *dst, int pitch, float *dst32, int *inexact,
                            hipStream_t stream);
hipError_t launch_make_poses(const double *qt, int n, int count, const ProblemDesc *probs, const GroupDesc *groups,
                             PoseState *out, hipStream_t stream);
hipError_t launch_grid_to_image(int dtype, const double *grid, int W, int H, void *dst, int pitch, float *dst32, int *inexact,
                                hipStream_t stream);
hipError_t launch_aos_to_soa(int dtype, const double *src, long long n, int stride, void *x, void *y, void *z, hipStream_t stream);
hipError_t launch_selftest_reduce(const float *in, float *a, float *b, float *c, float *d, double *o32, double *o64,
                                  hipStream_t stream);
// ea_preprocess.hip
hipError_t launch_resize_half_bgr8(const uint8_t *src, int H, int W, uint8_t *dst, hipStream_t s);
hipError_t launch_resize_half_f32(const float *src, int H, int W, float *dst, int nan_to_zero, hipStream_t s);
hipError_t launch_nan_to_zero(float *img, size_t n, hipStream_t s);
hipError_t launch_edge_strength(const uint8_t *bgr, int H, int W, uint8_t *gray, uint8_t *lap, hipStream_t s);
hipError_t launch_threshold_median(const uint8_t *lap, int H, int W, int thr, int median, uint8_t *mask, hipStream_t s);
hipError_t launch_chamfer(const uint8_t *mask, int H, int W, int *G, int *scratch, int *dist_fix, float *dist_f32,
                          unsigned int *minmax, hipStream_t s);
hipError_t launch_canny(const uint8_t *bgr, int H, int W, int low, int high, int l2_bgr, const uint8_t *keep, uint8_t *gray,
                        int *mag, uint8_t *dir, uint8_t *label, uint8_t *edges, uint8_t *inv, int *changed, int *rounds_out,
                        hipStream_t s);
hipError_t launch_edge_scatter_ros(int dtype, const uint8_t *edges, const float *depth, int H, int W, const int *block_offsets,
                                   double fx, double fy, double cx, double cy, void *X, void *Y, void *Z, int capacity,
                                   hipStream_t s);
hipError_t launch_dt_store(int dtype, const int *dist_fix, const float *dist_f32, int H, int W, const unsigned int *minmax,
                           int normalize, double lo, double hi, void *dst, int pitch, float *plain, float *dst32, hipStream_t s);
hipError_t launch_gate_by_mask(uint8_t *grad, const uint8_t *mask, int H, int W, hipStream_t s);
hipError_t launch_edge_count_scan(const uint8_t *lap, const uint16_t *depth, int H, int W, int thr, int *block_counts,
                                  int *total, hipStream_t s);
hipError_t launch_edge_scatter(int dtype, const uint8_t *lap, const uint16_t *depth, int H, int W, int thr,
                               const int *block_offsets, double fx, double fy, double cx, double cy, double z_scaling,
                               void *X, void *Y, void *Z, int capacity, hipStream_t s);
}  // namespace ea

using namespace ea;

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

// for the library's other translation units (ea_comm.hip): the thread-local error message, a batch's stream
extern "C" int ea_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(e_ == hipErrorNoDevice ? EA_ERR_NO_DEVICE : EA_ERR_HIP,                     \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

// scope guards for the temporaries of the measurement / test hooks, so that an early HIPCHK return frees them
namespace {
hipError_t cached_malloc(void **out, size_t bytes, int device);  // (the resource cache below)
void cached_free(void *p);
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) cached_free(p); }  // (blocks that did not come from the cache fall through to hipFree)
  template <typename U> U *as() const { return static_cast<U *>(p); }
};
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};
struct EventList {
  std::vector<hipEvent_t> ev;
  ~EventList() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
};
}  // namespace


// ---- resource cache -----------------------------------------------------------------------------------------------
// A fresh problem's first solve used to cost 6 ms against 0.14 ms for the solve itself (scripts/archive/upload_probe.py): three
// hipMalloc for the points, one for the image, four hipMalloc + four hipHostMalloc + a stream for its batch, and the same
// number of frees -- each a driver call of 0.1-1 ms, hipFree also a device synchronisation.  That is what the ceres facade
// pays per ceres::Solve (one ea_problem per ceres::Problem, standalone_edge_align.cpp:256-293).  Freed device blocks,
// pinned host blocks and streams are therefore kept and handed out again: device blocks to any request they fit without
// wasting more than half, pinned blocks to requests of exactly their size and flags, streams to the next batch of the
// device.  Bounded (kCacheMaxBytes per kind, kCacheMaxEntries blocks); ea_release_cached_memory() returns everything
// to the driver.  hipFree used to synchronise the device implicitly; the cache keeps that guarantee explicitly: a recycled
// DEVICE block is handed out only after a device-wide drain that happened AFTER it was freed (one hipDeviceSynchronize --
// microseconds on a device that is idle or only holds the early-exit launches a finished solve left queued -- before the
// first reuse following any free; `stale`), so no launch enqueued on behalf of the previous owner can still touch it.
namespace {
struct CachedBlock { void *p; size_t bytes; int device; unsigned flags; };
struct ResourceCache {
  std::mutex mu;
  std::vector<CachedBlock> dev, pinned;
  std::vector<std::pair<int, hipStream_t>> streams;
  std::unordered_map<void *, CachedBlock> live;  // what is handed out (size / device / flags of a pointer)
  size_t dev_bytes = 0, pinned_bytes = 0;
  std::unordered_map<int, bool> stale;           // device -> a block was freed since the device was last drained
};
ResourceCache &cache() { static ResourceCache *c = new ResourceCache; return *c; }  // (never destroyed: frees at exit race the runtime's teardown)
constexpr size_t kCacheMaxBytes = (size_t)1 << 30;
constexpr size_t kCacheMaxEntries = 256;

hipError_t cached_malloc(void **out, size_t bytes, int device) {
  *out = nullptr;
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  ResourceCache &c = cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < c.dev.size(); ++i)
      if (c.dev[i].device == device && c.dev[i].bytes >= bytes && c.dev[i].bytes <= 2 * bytes + 4096 &&
          (best == (size_t)-1 || c.dev[i].bytes < c.dev[best].bytes))
        best = i;
    if (best != (size_t)-1) {
      CachedBlock blk = c.dev[best];
      c.dev.erase(c.dev.begin() + (long)best);
      c.dev_bytes -= blk.bytes;
      c.live[blk.p] = blk;
      *out = blk.p;
      bool &stale = c.stale[device];
      if (stale) {
        // launches queued for the block's previous owner (on any stream) finish before the new owner sees it
        const hipError_t es = hipDeviceSynchronize();
        if (es != hipSuccess) return es;
        stale = false;
      }
      return hipSuccess;
    }
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {  // the cache may be what is in the way: give it back and try once more
    (void)hipGetLastError();
    extern void release_cached_memory_locked_free();
    release_cached_memory_locked_free();
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> g(c.mu);
  c.live[p] = CachedBlock{p, bytes, device, 0u};
  *out = p;
  return hipSuccess;
}

void cached_free(void *p) {
  if (!p) return;
  ResourceCache &c = cache();
  CachedBlock blk{p, 0, -1, 0u};
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      blk = it->second;
      c.live.erase(it);
      if (c.dev_bytes + blk.bytes <= kCacheMaxBytes && c.dev.size() < kCacheMaxEntries) {
        c.dev.push_back(blk);
        c.dev_bytes += blk.bytes;
        c.stale[blk.device] = true;
        return;
      }
    }
  }
  (void)hipFree(p);
}

hipError_t cached_host_malloc(void **out, size_t bytes, unsigned flags, int device) {
  *out = nullptr;
  if (bytes == 0) bytes = 64;
  ResourceCache &c = cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    for (size_t i = 0; i < c.pinned.size(); ++i)
      if (c.pinned[i].bytes == bytes && c.pinned[i].flags == flags && c.pinned[i].device == device) {
        CachedBlock blk = c.pinned[i];
        c.pinned.erase(c.pinned.begin() + (long)i);
        c.pinned_bytes -= blk.bytes;
        c.live[blk.p] = blk;
        *out = blk.p;
        return hipSuccess;
      }
  }
  void *p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes, flags);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(c.mu);
  c.live[p] = CachedBlock{p, bytes, device, flags};
  *out = p;
  return hipSuccess;
}

void cached_host_free(void *p) {
  if (!p) return;
  ResourceCache &c = cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      const CachedBlock blk = it->second;
      c.live.erase(it);
      if (c.pinned_bytes + blk.bytes <= kCacheMaxBytes / 4 && c.pinned.size() < kCacheMaxEntries) {
        c.pinned.push_back(blk);
        c.pinned_bytes += blk.bytes;
        return;
      }
    }
  }
  (void)hipHostFree(p);
}

hipError_t cached_stream_create(hipStream_t *out, int device) {
  ResourceCache &c = cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    for (size_t i = 0; i < c.streams.size(); ++i)
      if (c.streams[i].first == device) {
        *out = c.streams[i].second;
        c.streams.erase(c.streams.begin() + (long)i);
        return hipSuccess;
      }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void cached_stream_destroy(hipStream_t s, int device) {  // (the caller has synchronised it)
  if (!s) return;
  ResourceCache &c = cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    if (c.streams.size() < 64) { c.streams.emplace_back(device, s); return; }
  }
  (void)hipStreamDestroy(s);
}

void release_cached_memory_locked_free() {
  ResourceCache &c = cache();
  std::vector<CachedBlock> dev, pinned;
  std::vector<std::pair<int, hipStream_t>> streams;
  {
    std::lock_guard<std::mutex> g(c.mu);
    dev.swap(c.dev); pinned.swap(c.pinned); streams.swap(c.streams);
    c.dev_bytes = c.pinned_bytes = 0;
  }
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) cur = -1;
  for (const CachedBlock &b : dev) { (void)hipSetDevice(b.device); (void)hipFree(b.p); }
  for (const CachedBlock &b : pinned) (void)hipHostFree(b.p);
  for (auto &st : streams) { (void)hipSetDevice(st.first); (void)hipStreamDestroy(st.second); }
  if (cur >= 0) (void)hipSetDevice(cur);
  (void)hipGetLastError();
}
}  // namespace

extern "C" int ea_release_cached_memory(void) {
  release_cached_memory_locked_free();
  return EA_OK;
}

struct ea_problem {
  int device = 0;
  int dtype = EA_F64;
  ea_camera cam{};
  int loss_kind = EA_LOSS_CAUCHY;  // the reference's `new CauchyLoss(1.)`
  double loss_a = 1.0;
  double z_guard = 0.01, z_eps = 0.0;
  int rot_transposed = 0;
  // residual variants (utils.h:102-421)
  int variant = 0;                       // bit 0 distortion, bit 1 second camera
  double dist[5] = {0, 0, 0, 0, 0};      // k1, k2, p1, p2, k3
  double T12[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double T12inv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<ea_problem *> terms;       // further residual families sharing this problem's pose
  int64_t n = 0;
  void *d_x = nullptr, *d_y = nullptr, *d_z = nullptr;
  bool own_points = false;
  // storage order of the points in HBM (ea_problem_set_point_order): order[i] = caller's index of stored point i;
  // empty = the caller's order
  int order_tile = -1, order_tile_used = 0;
  std::vector<int32_t> order;
  void *d_dt = nullptr;
  size_t dt_cap = 0;   // bytes allocated behind d_dt: a frame of the same size reuses the allocation
  // fp64 problems: the float32 mirror of the image (same padded layout and pitch in texels) and whether it holds every
  // value exactly -- then the plain fp64 kernels read it instead (ProblemDesc::dt32): one 16-byte load per stencil row
  float *d_dt32 = nullptr;
  size_t dt32_cap = 0;
  bool dt32_exact = false;
  size_t pts_cap = 0;  // bytes allocated behind each of d_x, d_y, d_z when own_points (hipFree / hipMalloc per frame
                       // cost more than the whole pre-processing of a 640x480 frame)
  int W = 0, H = 0, pitch = 0;
  uint64_t version = 1;  // bumped by every setter; batches rebuild their descriptors lazily
  ea_batch *self = nullptr;
  hipStream_t stream = nullptr;
  // scratch for the frame pre-processing kernels (grown on demand, reused across frames)
  unsigned char *ws = nullptr;
  size_t ws_bytes = 0;
  // full-resolution frames on their way to a half-resolution level (ea_problem_set_*_frame_ros_scaled)
  unsigned char *stage = nullptr;
  size_t stage_bytes = 0;
  // what the last producer call left in the workspace: 1 = set_now_frame (Laplacian strength), 2 = set_now_frame_canny
  // (edge map, no mask), 0 = nothing reusable; with the frame's extent.  The tracker extracts the same frame's edge
  // points from it instead of uploading and filtering the frame a second time.
  int ws_now_kind = 0, ws_now_h = 0, ws_now_w = 0;
};

struct ea_batch {
  std::vector<ea_problem *> probs;
  std::vector<uint64_t> versions;
  int device = 0, dtype = EA_F64;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // device
  // descriptors on the device: ONE block [ProblemDesc x nterms | GroupDesc x count], the layout of the staging block, so that a
  // build is one upload (a frame pair that changes its points / image every solve pays it every solve)
  unsigned char *d_desc_block = nullptr;
  ProblemDesc *d_probs = nullptr;       // one per term (problem + its additional terms), groups contiguous
  GroupDesc *d_groups = nullptr;         // one per problem (= pose); behind the terms of the current build
  // descriptors go up from a pinned staging block with asynchronous copies on the batch's stream (a frame pair that
  // changes its points / DT every solve would otherwise pay two blocking copies per solve)
  unsigned char *h_desc = nullptr;
  size_t h_desc_cap = 0;
  hipEvent_t desc_done = nullptr;
  int nterms = 0, terms_cap = 0;
  int any_variant = 0, terms_are_groups = 1;
  int ntiles = 0, tiles_cap = 0;  // rows of the partial-sum array (one per workgroup with work)
  int chunk = 256, max_chunks = 0; // points per workgroup; largest per-problem workgroup count
  PoseState *d_poses = nullptr;
  double *d_partials = nullptr;
  EvalOut *d_out = nullptr;
  unsigned char *d_lm_block = nullptr;  // [PoseState x count | LMState x count | LMTrace x count]: poses + states go up
                                        // in ONE copy per solve, states + traces come back in ONE copy
  LMState *d_states = nullptr;
  LMCold *d_cold = nullptr;             // written by the first LM step before anything reads it
  LMTrace *d_traces = nullptr;
  int *d_progress = nullptr;            // device view of h_progress
  // pinned host mirrors
  PoseState *h_poses = nullptr;
  PoseState *dv_h_poses = nullptr;      // the device's view of h_poses: a lone evaluation of a few problems reads its poses from there
  int t_zero_copy = -1;                 // tuning key "zero_copy_poses": 0 = always upload the poses first (A/B)
  EvalOut *h_out = nullptr;
  EvalOut *dv_out = nullptr;            // the device's view of h_out: the synchronous evaluations fold straight into host memory
  // final delivery of a solve: [LMState x count | LMTrace x count] in pinned, device-mapped host memory, written by the
  // step kernel that ends a problem's solve (dv_* = the device's view of the same memory)
  unsigned char *h_deliver = nullptr;
  LMState *hd_states = nullptr, *dv_states = nullptr;
  LMTrace *hd_traces = nullptr, *dv_traces = nullptr;
  unsigned char *h_lm_block = nullptr;
  LMState *h_states = nullptr;
  LMTrace *h_traces = nullptr;
  int *h_progress = nullptr;            // pinned, device-visible: [running x count | evaluations started x count | steps complete x count | done flag]
  unsigned int *d_done_count = nullptr; // workgroups of the last fold of a synchronous evaluation that have delivered (ea_reduce_done_kernel)
  int done_seq = 0;                     // the value the flag takes when the current call's results have all landed
  int t_poll = 1;                       // tuning key "poll_results": 0 = wait for the stream's completion signal instead (A/B)
  // tuning (-1 = heuristic)
  int t_lds_bytes = -1, t_ppt = -1, t_use_lds = -1, t_xcd = -1, t_nt = -1, t_streams = -1, t_variant = 0;
  int t_wide = 0, wide = 0;              // "wide_accumulate": an fp32 kernel sums in fp64 from the lane's sum on (plain functor, L2 path)
  int ppt = 1, nt = 256, lds_bytes = 0, xcd_remap = 1;
  int t_test_fail_build = 0;  // test hook: the next descriptor build fails half-way, as a failed allocation would
  int t_test_stall_ms = 0;  // test hook: hold the stream on a host function for this long at the start of a solve
  int t_buf = -1, buffer_loads = 0;  // raw-buffer addressing of the DT image and the points (needs a < 2 GiB image)
  int t_img32 = -1, img32 = 0;       // fp64 batch whose kernels read the float32 mirror of the DT images ("dt_f32": -1 = when every
                                     // term has an exact mirror, 0 = never)
  std::vector<ea_batch *> parts;  // sub-batches of the concurrent solve (ea_batch_solve)
  bool built = false;
  hipEvent_t bench_e0 = nullptr, bench_e1 = nullptr;
  hipGraphExec_t bench_graph = nullptr;  // K x (evaluation + fold) captured once (ea_batch_bench_capture), replayed by
  int bench_riding_steps = 0;            // length of the last riding-fold sequence (captured or run launch by launch)
  int bench_graph_steps = 0;             // ea_batch_bench_steps(K): the timed region then holds no per-launch host work
  // ea_batch_bench_capture_pipelined: the fold of step k-1 rides in the launch of evaluation k; the evaluations alternate
  // between `bench_ring` = 2 row / result arrays
  int bench_ring = 0;
  double *d_bench_rows = nullptr;   // bench_ring x tiles_cap x kAccSlots
  EvalOut *d_bench_out = nullptr;   // bench_ring x count
  // ea_batch_set_poses / ea_batch_eval_resident_poses: K different poses per problem resident on the device, their K
  // evaluations + folds as one replayed hipGraph (the fold of evaluation k-1 rides in the launch of evaluation k), the K
  // results folded straight into pinned host memory
  int kp_K = 0, kp_cap = 0;                       // poses per problem resident / allocated
  double *h_kqt = nullptr, *d_kqt = nullptr;      // q, t staging: cap x count x 7 doubles (pinned / device)
  PoseState *d_kposes = nullptr;                  // cap x count
  EvalOut *h_kout = nullptr, *dv_kout = nullptr;  // cap x count, pinned + the device's view of it
  // G poses per launch: the descriptor / group tables replicated G times (copy g of a term owns the partial rows and the pose
  // slot of pose g) and the partial rows of G poses
  int kp_G = 0, t_kp_G = 0;   // (t_kp_G > 0: tuning key "poses_per_launch" caps G)
  // the launch shape of the pose-batched evaluation -- a throughput shape (more points per lane than the latency shape one
  // evaluation of the same batch takes) -- and the partial rows / widest term it implies per pose
  int kp_ppt = 1, kp_nt = 256, kp_chunk = 256, kp_ntiles = 0, kp_max_chunks = 0;
  ProblemDesc *d_kprobs = nullptr;
  GroupDesc *d_kgroups = nullptr;
  double *d_krows = nullptr;
  // one launch per LM iteration (ea_lm_iter_kernel): the second buffer of each pair a launch reads / writes -- states and cold
  // systems [LMState x count | LMCold x count] and partial rows -- beside d_states, d_cold, d_partials
  unsigned char *d_iter_alt = nullptr;
  int iter_alt_count = 0;
  double *d_partials_alt = nullptr;
  int tiles_cap_alt = 0;
  int t_fused = -1;             // tuning key "fused_iterations": -1 = whenever the solve qualifies, 0 = never (A/B, tests)
  int last_fused = 0;           // info key "fused_iterations": the last solve of this batch ran one launch per iteration
  bool needs_drain = false;     // a solve gave up on its deadline with launches still queued: synchronise before reuse
  const void *x0 = nullptr, *y0 = nullptr, *z0 = nullptr;  // problem 0's point arrays and count, handed to the evaluation
  int n0 = 0;                                              // kernel in its preloaded arguments
  GroupDesc group0 = {0, 0, 0, 0};  // host copy of d_groups[0]: handed to the step kernel by value
  GroupDesc *d_one_row = nullptr;  // {0, 1, 0, 1}: "one partial row" for the step kernel of ea_solve_sharded_device
  bool poses_uploaded = false;  // d_poses holds caller-supplied poses (ea_batch_bench_steps re-evaluates at them)
  bool poses_staged = false;    // h_poses holds the poses of the last ea_batch_eval, which the kernel read in place (not in d_poses yet)
  // materialised mode (ea_batch_eval_rows*): rows of all terms, in term order; library-owned output arrays on request
  int64_t total_rows = 0, max_n = 0;
  std::vector<int64_t> row_offsets;     // per problem (its terms are adjacent), count + 1 entries
  void *d_rows_r = nullptr, *d_rows_J = nullptr;
  int64_t rows_cap = 0;
  unsigned int *d_rows_invalid = nullptr;
  int t_rows_staged = -1, t_rows_nt = -1;
};

static int check_device(int device) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    return fail(EA_ERR_NO_DEVICE, "no HIP device available (libea_hip has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(EA_ERR_INVALID_ARG, "device index out of range");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(EA_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  return EA_OK;
}

extern "C" const char *ea_last_error(void) { return g_err.c_str(); }

extern "C" void *ea_host_alloc(size_t bytes, int device) {
  if (bytes == 0) { fail(EA_ERR_INVALID_ARG, "ea_host_alloc: zero bytes"); return nullptr; }
  if (check_device(device) != EA_OK) return nullptr;
  void *p = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
  if (e != hipSuccess) { fail(EA_ERR_ALLOC, std::string("ea_host_alloc: ") + hipGetErrorString(e)); return nullptr; }
  return p;
}

extern "C" void ea_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}
extern "C" const char *ea_version(void) { return "edge_alignment_amd 0.2 (gfx950)"; }

extern "C" int ea_device_count(int *count) {
  if (!count) return fail(EA_ERR_INVALID_ARG, "count is NULL");
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess) { *count = 0; return fail(EA_ERR_NO_DEVICE, hipGetErrorString(e)); }
  int n950 = 0;
  for (int i = 0; i < cnt; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++n950;
  }
  *count = n950;
  return EA_OK;
}

extern "C" void ea_default_options(ea_options *o) {
  if (!o) return;
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
  o->strategy = EA_STRATEGY_LM;
  o->minimizer_progress_to_stdout = 0;
  o->iterations_per_sync = 0;
  o->solve_timeout_ms = 0.0;
}

// ceres::Solver::Options::IsValid for the fields carried here (solver.cc: OPTION_GE / OPTION_GT / OPTION_LE_OPTION);
// written so that a NaN fails every test.  Ceres ends such a Solve with FAILURE before touching the problem; the C-ABI
// returns EA_ERR_INVALID_ARG and leaves q, t and the summary alone.
static int check_options(const ea_options &o) {
  if (!(o.max_num_iterations >= 0)) return fail(EA_ERR_INVALID_ARG, "max_num_iterations < 0");
  if (o.strategy != EA_STRATEGY_LM && o.strategy != EA_STRATEGY_DOGLEG)
    return fail(EA_ERR_INVALID_ARG, "unknown trust-region strategy");
  if (!(o.function_tolerance >= 0.0) || !(o.gradient_tolerance >= 0.0) || !(o.parameter_tolerance >= 0.0))
    return fail(EA_ERR_INVALID_ARG, "tolerances must be >= 0");
  if (!(o.min_trust_region_radius > 0.0) || !(o.initial_trust_region_radius > 0.0) || !(o.max_trust_region_radius > 0.0) ||
      !(o.min_trust_region_radius <= o.initial_trust_region_radius) || !(o.initial_trust_region_radius <= o.max_trust_region_radius))
    return fail(EA_ERR_INVALID_ARG, "need 0 < min_trust_region_radius <= initial_trust_region_radius <= max_trust_region_radius");
  if (!(o.min_relative_decrease >= 0.0)) return fail(EA_ERR_INVALID_ARG, "min_relative_decrease < 0");
  if (!(o.min_lm_diagonal >= 0.0) || !(o.min_lm_diagonal <= o.max_lm_diagonal))
    return fail(EA_ERR_INVALID_ARG, "need 0 <= min_lm_diagonal <= max_lm_diagonal");
  if (!(o.max_num_consecutive_invalid_steps >= 0)) return fail(EA_ERR_INVALID_ARG, "max_num_consecutive_invalid_steps < 0");
  if (o.iterations_per_sync < 0) return fail(EA_ERR_INVALID_ARG, "iterations_per_sync < 0");
  if (o.solve_timeout_ms != o.solve_timeout_ms) return fail(EA_ERR_INVALID_ARG, "solve_timeout_ms is NaN");
  return EA_OK;
}

// ---- problem ------------------------------------------------------------------------------------

extern "C" int ea_problem_create(ea_problem **out, const ea_camera *cam, int dtype, int device) {
  if (!out || !cam) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (dtype != EA_F64 && dtype != EA_F32) return fail(EA_ERR_INVALID_ARG, "dtype must be EA_F64 or EA_F32");
  if (!(std::fabs(cam->fx) <= DBL_MAX) || !(std::fabs(cam->fy) <= DBL_MAX) || !(std::fabs(cam->cx) <= DBL_MAX) ||
      !(std::fabs(cam->cy) <= DBL_MAX) || cam->fx == 0.0 || cam->fy == 0.0)
    return fail(EA_ERR_INVALID_ARG, "camera intrinsics must be finite, focal lengths non-zero");
  int rc = check_device(device);
  if (rc != EA_OK) return rc;
  ea_problem *p = new (std::nothrow) ea_problem();
  if (!p) return fail(EA_ERR_ALLOC, "out of host memory");
  p->device = device;
  p->dtype = dtype;
  p->cam = *cam;
  *out = p;
  return EA_OK;
}

static void free_points(ea_problem *p) {
  if (p->own_points) {
    cached_free(p->d_x); cached_free(p->d_y); cached_free(p->d_z);
  }
  p->d_x = p->d_y = p->d_z = nullptr;
  p->own_points = false;
  p->pts_cap = 0;
  p->n = 0;
  p->order.clear();
  p->order_tile_used = 0;
}

// Room for n points in arrays the problem owns: the previous allocation when it is large enough (and not more than
// four times too large), a fresh one otherwise.  Leaves the problem without points (n = 0) either way.
static int reserve_points(ea_problem *p, int64_t n) {
  const size_t need = (size_t)n * (p->dtype == EA_F32 ? 4 : 8);
  if (p->own_points && p->pts_cap >= need && p->pts_cap <= 4 * need + 4096) {
    p->n = 0;
    p->order.clear();
    p->order_tile_used = 0;
    return EA_OK;
  }
  free_points(p);
  if (n == 0) return EA_OK;
  p->own_points = true;  // (before the allocations: a failure half-way leaves what was allocated to free_points)
  HIPCHK(cached_malloc(&p->d_x, need, p->device));
  HIPCHK(cached_malloc(&p->d_y, need, p->device));
  HIPCHK(cached_malloc(&p->d_z, need, p->device));
  p->pts_cap = need;
  return EA_OK;
}

// Storage order for large point sets: tiles of T x T pixels of the reference frame (the identity-pose projection),
// tiles in raster order, the caller's order inside a tile.  A wavefront's 64 points then sample a compact patch of
// the DT image instead of a 1-2 row strip across its whole width, and the four stencil-row loads of neighbouring
// points hit the lines the CU's L1 already holds (C5: 12.7 -> 8.1 us per evaluation, scripts/archive/order_sweep.py).
// Sums are taken in storage order; per-point outputs and ea_problem_get_points stay in the caller's order.
constexpr int64_t kAutoOrderPoints = 200000;  // below this a launch is latency-bound and the order does not matter

static void tile_order(const ea_problem *p, const double *xyz, int64_t n, int64_t stride, int tile,
                       std::vector<int32_t> &order) {
  std::vector<uint32_t> key((size_t)n);
  uint32_t max_tx = 0, max_ty = 0;
  const double inv = 1.0 / (double)tile;
  for (int64_t i = 0; i < n; ++i) {
    const double x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    double u = p->cam.fx * x / z + p->cam.cx, v = p->cam.fy * y / z + p->cam.cy;
    if (!(u >= 0.0)) u = 0.0;  // also NaN (z = 0)
    if (!(v >= 0.0)) v = 0.0;
    const uint32_t tx = (uint32_t)std::min(u * inv, 4095.0), ty = (uint32_t)std::min(v * inv, 4095.0);
    key[(size_t)i] = (ty << 12) | tx;
    max_tx = std::max(max_tx, tx); max_ty = std::max(max_ty, ty);
  }
  // stable counting sort on (ty, tx)
  const uint32_t nx = max_tx + 1, ny = max_ty + 1;
  std::vector<int64_t> head((size_t)nx * ny + 1, 0);
  for (int64_t i = 0; i < n; ++i) { const uint32_t k = key[(size_t)i]; head[(size_t)(k >> 12) * nx + (k & 4095) + 1]++; }
  for (size_t c = 1; c < head.size(); ++c) head[c] += head[c - 1];
  order.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) { const uint32_t k = key[(size_t)i]; order[(size_t)head[(size_t)(k >> 12) * nx + (k & 4095)]++] = (int32_t)i; }
}

extern "C" void ea_batch_destroy(ea_batch *b);

extern "C" void ea_problem_destroy(ea_problem *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->self) ea_batch_destroy(p->self);
  free_points(p);
  if (p->d_dt) cached_free(p->d_dt);
  if (p->d_dt32) cached_free(p->d_dt32);
  if (p->ws) cached_free(p->ws);
  if (p->stage) cached_free(p->stage);
  delete p;
}

extern "C" int64_t ea_problem_num_points(const ea_problem *p) { return p ? p->n : 0; }

extern "C" int ea_problem_set_distortion(ea_problem *p, double k1, double k2, double p1, double p2, double k3) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!(std::fabs(k1) <= DBL_MAX) || !(std::fabs(k2) <= DBL_MAX) || !(std::fabs(p1) <= DBL_MAX) || !(std::fabs(p2) <= DBL_MAX) ||
      !(std::fabs(k3) <= DBL_MAX))
    return fail(EA_ERR_INVALID_ARG, "distortion coefficients must be finite");
  p->dist[0] = k1; p->dist[1] = k2; p->dist[2] = p1; p->dist[3] = p2; p->dist[4] = k3;
  const bool on = k1 != 0.0 || k2 != 0.0 || p1 != 0.0 || p2 != 0.0 || k3 != 0.0;
  p->variant = on ? (p->variant | 1) : (p->variant & ~1);
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_set_second_camera(ea_problem *p, const double trans_1to2[16], const double trans_1to2_inv[16]) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!trans_1to2 || !trans_1to2_inv) {  // back to the first camera
    for (int i = 0; i < 16; ++i) p->T12[i] = p->T12inv[i] = (i % 5 == 0) ? 1.0 : 0.0;
    p->variant &= ~2;
    p->version++;
    return EA_OK;
  }
  for (int k = 0; k < 2; ++k) {
    const double *m = k ? trans_1to2_inv : trans_1to2;
    if (m[12] != 0.0 || m[13] != 0.0 || m[14] != 0.0 || m[15] != 1.0)
      return fail(EA_ERR_INVALID_ARG, "rig transforms must be affine (last row 0 0 0 1)");
    for (int i = 0; i < 12; ++i)
      if (!(std::fabs(m[i]) <= DBL_MAX)) return fail(EA_ERR_INVALID_ARG, "rig transforms must be finite");
  }
  for (int i = 0; i < 16; ++i) { p->T12[i] = trans_1to2[i]; p->T12inv[i] = trans_1to2_inv[i]; }
  p->variant |= 2;
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_add_term(ea_problem *p, ea_problem *term) {
  if (!p || !term || p == term) return fail(EA_ERR_INVALID_ARG, "bad argument");
  if (!term->terms.empty()) return fail(EA_ERR_INVALID_ARG, "a term cannot have terms of its own");
  if (p->device != term->device || p->dtype != term->dtype) return fail(EA_ERR_INVALID_ARG, "terms must share device and dtype");
  if (std::find(p->terms.begin(), p->terms.end(), term) != p->terms.end()) return fail(EA_ERR_INVALID_ARG, "term already added");
  p->terms.push_back(term);
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_clear_terms(ea_problem *p) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  p->terms.clear();
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_set_points(ea_problem *p, const double *xyz, int64_t n, int64_t stride) {
  if (!p || (n > 0 && !xyz)) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (n < 0 || n > 0x7fffff00LL) return fail(EA_ERR_INVALID_ARG, "n out of range");
  if (stride < 3) return fail(EA_ERR_INVALID_ARG, "stride_elems must be >= 3");
  HIPCHK(hipSetDevice(p->device));
  p->version++;
  if (n == 0) { free_points(p); return EA_OK; }
  const size_t esz = p->dtype == EA_F32 ? 4 : 8;
  const int tile = p->order_tile < 0 ? (n >= kAutoOrderPoints ? 16 : 0) : p->order_tile;
  if (tile == 0 && stride <= 8) {
    // the caller's order, compact AoS: the array goes up as it is (one copy) and is split into x[], y[], z[] of the
    // problem's dtype on the device -- no host pass over the points, one upload instead of three
    int rc = reserve_points(p, n);
    if (rc != EA_OK) return rc;
    const size_t raw = ((size_t)(n - 1) * (size_t)stride + 3) * sizeof(double);
    void *d_raw = nullptr;
    HIPCHK(cached_malloc(&d_raw, raw, p->device));
    hipError_t e = hipMemcpy(d_raw, xyz, raw, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_aos_to_soa(p->dtype, static_cast<const double *>(d_raw), n, (int)stride, p->d_x, p->d_y, p->d_z, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    cached_free(d_raw);
    if (e != hipSuccess) { free_points(p); return fail(EA_ERR_HIP, std::string("ea_problem_set_points: ") + hipGetErrorString(e)); }
    p->n = n;
    p->order.clear();
    p->order_tile_used = 0;
    return EA_OK;
  }
  std::vector<unsigned char> soa(3 * (size_t)n * esz);
  std::vector<int32_t> order;
  if (tile > 0) tile_order(p, xyz, n, stride, tile, order);
  const int32_t *ord = order.empty() ? nullptr : order.data();
  for (int c = 0; c < 3; ++c) {
    if (p->dtype == EA_F32) {
      float *dst = reinterpret_cast<float *>(soa.data()) + (size_t)c * n;
      for (int64_t i = 0; i < n; ++i) dst[i] = (float)xyz[(ord ? ord[i] : i) * stride + c];
    } else {
      double *dst = reinterpret_cast<double *>(soa.data()) + (size_t)c * n;
      for (int64_t i = 0; i < n; ++i) dst[i] = xyz[(ord ? ord[i] : i) * stride + c];
    }
  }
  {
    int rc = reserve_points(p, n);
    if (rc != EA_OK) return rc;
  }
  HIPCHK(hipMemcpy(p->d_x, soa.data(), n * esz, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->d_y, soa.data() + (size_t)n * esz, n * esz, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->d_z, soa.data() + 2 * (size_t)n * esz, n * esz, hipMemcpyHostToDevice));
  p->n = n;
  p->order.swap(order);
  p->order_tile_used = tile;
  return EA_OK;
}

extern "C" int ea_problem_get_point_order(const ea_problem *p, int *tile_px) {
  if (!p || !tile_px) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  *tile_px = p->order_tile_used;
  return EA_OK;
}

extern "C" int ea_problem_set_point_order(ea_problem *p, int tile_px) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (tile_px > 1024) return fail(EA_ERR_INVALID_ARG, "tile_px must be <= 1024 (0 = caller's order, < 0 = automatic)");
  p->order_tile = tile_px < 0 ? -1 : tile_px;
  return EA_OK;
}

extern "C" int ea_problem_set_points_device(ea_problem *p, const void *x, const void *y, const void *z, int64_t n) {
  if (!p || (n > 0 && (!x || !y || !z))) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (n < 0 || n > 0x7fffff00LL) return fail(EA_ERR_INVALID_ARG, "n out of range");
  HIPCHK(hipSetDevice(p->device));
  free_points(p);
  p->d_x = const_cast<void *>(x); p->d_y = const_cast<void *>(y); p->d_z = const_cast<void *>(z);
  p->own_points = false;
  p->n = n;
  p->version++;
  return EA_OK;
}

static int alloc_dt(ea_problem *p, int W, int H) {
  p->W = W; p->H = H;
  p->pitch = (W + 2 * kImagePad + 3) & ~3;
  const size_t esz = p->dtype == EA_F32 ? 4 : 8;
  const size_t need = (size_t)p->pitch * (size_t)(H + 2 * kImagePad) * esz;
  p->dt32_exact = false;  // (until the call that fills the image says otherwise)
  if (p->dtype == EA_F64) {
    const size_t need32 = need / 2;
    if (!(p->d_dt32 && p->dt32_cap >= need32 && p->dt32_cap <= 4 * need32)) {
      if (p->d_dt32) { cached_free(p->d_dt32); p->d_dt32 = nullptr; p->dt32_cap = 0; }
      HIPCHK(cached_malloc(reinterpret_cast<void **>(&p->d_dt32), need32, p->device));
      p->dt32_cap = need32;
    }
  }
  if (p->d_dt && p->dt_cap >= need && p->dt_cap <= 4 * need) return EA_OK;  // same-size frame: keep the allocation
  if (p->d_dt) { cached_free(p->d_dt); p->d_dt = nullptr; p->dt_cap = 0; }
  HIPCHK(cached_malloc(&p->d_dt, need, p->device));
  p->dt_cap = need;
  return EA_OK;
}

// a device word for the "mirror is not exact" flag of the two upload paths, zeroed; freed by the caller
static int inexact_flag(ea_problem *p, int **flag) {
  *flag = nullptr;
  if (p->dtype != EA_F64) return EA_OK;
  HIPCHK(cached_malloc(reinterpret_cast<void **>(flag), sizeof(int), p->device));
  HIPCHK(hipMemsetAsync(*flag, 0, sizeof(int), nullptr));
  return EA_OK;
}

extern "C" int ea_problem_set_dt(ea_problem *p, const double *data, int grid_rows, int grid_cols) {
  if (!p || !data) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (grid_rows < 1 || grid_cols < 1 || grid_rows > 32768 || grid_cols > 32768)
    return fail(EA_ERR_INVALID_ARG, "grid extent out of range");
  HIPCHK(hipSetDevice(p->device));
  // Grid2D rows index u, cols index v (ref: standalone_edge_align.cpp:258, utils.h:77).
  const int W = grid_rows, H = grid_cols;
  int rc = alloc_dt(p, W, H);
  if (rc != EA_OK) return rc;
  // The grid goes up as it is (one contiguous copy) and is transposed, bordered and converted on the device: the host loop
  // this replaces read the caller's array with a stride of one grid row per element (0.28 -> 0.1x ms per 640 x 480 frame).
  const size_t raw = (size_t)W * (size_t)H * sizeof(double);
  void *d_raw = nullptr;
  int *d_inexact = nullptr;
  if ((rc = inexact_flag(p, &d_inexact)) != EA_OK) return rc;
  hipError_t e = cached_malloc(&d_raw, raw, p->device);
  if (e == hipSuccess) e = hipMemcpy(d_raw, data, raw, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_grid_to_image(p->dtype, static_cast<const double *>(d_raw), W, H, p->d_dt, p->pitch, p->d_dt32, d_inexact, nullptr);
  int inexact = 1;
  if (e == hipSuccess && d_inexact) e = hipMemcpy(&inexact, d_inexact, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  cached_free(d_raw);
  cached_free(d_inexact);
  if (e != hipSuccess) return fail(EA_ERR_HIP, std::string("ea_problem_set_dt: ") + hipGetErrorString(e));
  // every grid the reference builds comes out of a CV_32F distance transform (utils.cpp:79-82, cv2eigen at
  // standalone_edge_align.cpp:205-206): its doubles are floats, and the fp64 kernels may read the float mirror
  p->dt32_exact = d_inexact != nullptr && inexact == 0;
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_set_dt_image_device(ea_problem *p, const void *image, int height, int width) {
  if (!p || !image) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (height < 1 || width < 1 || height > 32768 || width > 32768)
    return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  HIPCHK(hipSetDevice(p->device));
  int rc = alloc_dt(p, width, height);
  if (rc != EA_OK) return rc;
  int *d_inexact = nullptr;
  if ((rc = inexact_flag(p, &d_inexact)) != EA_OK) return rc;
  hipError_t e = launch_pad_image(p->dtype, image, height, width, p->d_dt, p->pitch, p->d_dt32, d_inexact, nullptr);
  int inexact = 1;
  if (e == hipSuccess && d_inexact) e = hipMemcpy(&inexact, d_inexact, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  cached_free(d_inexact);
  if (e != hipSuccess) return fail(EA_ERR_HIP, std::string("ea_problem_set_dt_image_device: ") + hipGetErrorString(e));
  p->dt32_exact = d_inexact != nullptr && inexact == 0;
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_set_loss(ea_problem *p, int kind, double a) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (kind < EA_LOSS_TRIVIAL || kind > EA_LOSS_HUBER) return fail(EA_ERR_INVALID_ARG, "unknown loss kind");
  if (kind != EA_LOSS_TRIVIAL && !(a > 0.0)) return fail(EA_ERR_INVALID_ARG, "loss scale must be > 0");
  p->loss_kind = kind;
  p->loss_a = a;
  p->version++;
  return EA_OK;
}

extern "C" int ea_problem_set_flavour(ea_problem *p, double z_guard, double z_eps, int rot_transposed) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!(z_guard >= 0.0) || !(z_guard <= DBL_MAX) || !(std::fabs(z_eps) <= DBL_MAX))
    return fail(EA_ERR_INVALID_ARG, "z_guard must be >= 0, z_guard and z_eps finite");
  p->z_guard = z_guard; p->z_eps = z_eps; p->rot_transposed = rot_transposed ? 1 : 0;
  p->version++;
  return EA_OK;
}

// ---- batch --------------------------------------------------------------------------------------

static void bench_ring_free(ea_batch *b) {
  (void)hipFree(b->d_bench_rows); (void)hipFree(b->d_bench_out);
  b->d_bench_rows = nullptr; b->d_bench_out = nullptr; b->bench_ring = 0; b->bench_riding_steps = 0;
}

static void kposes_free_tables(ea_batch *b);

static void kposes_free(ea_batch *b) {
  kposes_free_tables(b);
  cached_free(b->d_kqt); cached_free(b->d_kposes);
  cached_host_free(b->h_kqt); cached_host_free(b->h_kout);
  b->d_kqt = nullptr; b->d_kposes = nullptr; b->h_kqt = nullptr; b->h_kout = nullptr; b->dv_kout = nullptr;
  b->kp_K = b->kp_cap = 0;
}

static void batch_free_device(ea_batch *b) {
  kposes_free(b);
  (void)hipFree(b->d_rows_r); (void)hipFree(b->d_rows_J); (void)hipFree(b->d_rows_invalid);
  b->d_rows_r = b->d_rows_J = nullptr; b->d_rows_invalid = nullptr; b->rows_cap = 0;
  if (b->bench_graph) { (void)hipGraphExecDestroy(b->bench_graph); b->bench_graph = nullptr; b->bench_graph_steps = 0; b->bench_riding_steps = 0; }
  bench_ring_free(b);
  if (b->bench_e0) { (void)hipEventDestroy(b->bench_e0); b->bench_e0 = nullptr; }
  if (b->bench_e1) { (void)hipEventDestroy(b->bench_e1); b->bench_e1 = nullptr; }
  (void)hipFree(b->d_one_row); b->d_one_row = nullptr;
  cached_free(b->d_desc_block); cached_free(b->d_lm_block); cached_free(b->d_cold);
  b->d_desc_block = nullptr;
  cached_free(b->d_iter_alt); cached_free(b->d_partials_alt);
  b->d_iter_alt = nullptr; b->d_partials_alt = nullptr; b->iter_alt_count = 0; b->tiles_cap_alt = 0;
  cached_free(b->d_partials); cached_free(b->d_out); cached_free(b->d_done_count);
  b->d_done_count = nullptr;
  cached_host_free(b->h_lm_block); cached_host_free(b->h_out);
  cached_host_free(b->h_progress); cached_host_free(b->h_deliver);
  if (b->h_desc) cached_host_free(b->h_desc);
  if (b->desc_done) (void)hipEventDestroy(b->desc_done);
  b->h_desc = nullptr; b->h_desc_cap = 0; b->desc_done = nullptr;
  b->d_probs = nullptr; b->d_groups = nullptr; b->d_poses = nullptr; b->d_partials = nullptr; b->d_traces = nullptr; b->d_cold = nullptr; b->d_lm_block = nullptr;
  b->d_out = nullptr; b->d_states = nullptr; b->d_progress = nullptr;
  b->h_poses = nullptr; b->h_out = nullptr; b->dv_out = nullptr; b->h_states = nullptr; b->h_traces = nullptr; b->h_progress = nullptr;
  b->h_lm_block = nullptr; b->h_deliver = nullptr;
}

extern "C" int ea_batch_create(ea_batch **out, ea_problem *const *problems, int count) {
  if (!out || !problems || count < 1) return fail(EA_ERR_INVALID_ARG, "need at least one problem");
  for (int i = 0; i < count; ++i) {
    if (!problems[i]) return fail(EA_ERR_INVALID_ARG, "NULL problem in batch");
    if (problems[i]->device != problems[0]->device || problems[i]->dtype != problems[0]->dtype)
      return fail(EA_ERR_INVALID_ARG, "all problems of a batch must share device and dtype");
  }
  ea_batch *b = new (std::nothrow) ea_batch();
  if (!b) return fail(EA_ERR_ALLOC, "out of host memory");
  b->probs.assign(problems, problems + count);
  b->versions.assign(count, 0);
  b->device = problems[0]->device;
  b->dtype = problems[0]->dtype;
  hipError_t e = hipSetDevice(b->device);
  if (e == hipSuccess) e = cached_stream_create(&b->stream, b->device);
  if (e != hipSuccess) { delete b; return fail(EA_ERR_HIP, hipGetErrorString(e)); }
  b->own_stream = true;
  const size_t c = (size_t)count;
  const size_t lm_bytes = c * (sizeof(LMState) + sizeof(LMTrace) + sizeof(PoseState));
  e = cached_malloc(reinterpret_cast<void **>(&b->d_lm_block), lm_bytes, b->device);
  if (e == hipSuccess) e = cached_malloc(reinterpret_cast<void **>(&b->d_out), c * sizeof(EvalOut), b->device);
  if (e == hipSuccess) e = cached_malloc(reinterpret_cast<void **>(&b->d_cold), c * sizeof(LMCold), b->device);
  if (e == hipSuccess) e = cached_host_malloc(reinterpret_cast<void **>(&b->h_lm_block), lm_bytes, hipHostMallocDefault, b->device);
  if (e == hipSuccess) e = cached_host_malloc(reinterpret_cast<void **>(&b->h_out), c * sizeof(EvalOut), hipHostMallocMapped, b->device);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&b->dv_out), b->h_out, 0);
  if (e == hipSuccess) e = cached_host_malloc(reinterpret_cast<void **>(&b->h_progress), (3 * c + 1) * sizeof(int), hipHostMallocMapped, b->device);
  if (e == hipSuccess) e = cached_malloc(reinterpret_cast<void **>(&b->d_done_count), 256, b->device);
  if (e == hipSuccess) e = hipMemsetAsync(b->d_done_count, 0, 256, b->stream);
  if (e == hipSuccess) b->h_progress[3 * c] = 0;
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_progress), b->h_progress, 0);
  if (e == hipSuccess) e = cached_host_malloc(reinterpret_cast<void **>(&b->h_deliver), c * (sizeof(LMState) + sizeof(LMTrace)), hipHostMallocMapped, b->device);
  if (e == hipSuccess) {
    unsigned char *dv = nullptr;
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dv), b->h_deliver, 0);
    b->hd_states = reinterpret_cast<LMState *>(b->h_deliver);
    b->hd_traces = reinterpret_cast<LMTrace *>(b->h_deliver + c * sizeof(LMState));
    b->dv_states = reinterpret_cast<LMState *>(dv);
    b->dv_traces = reinterpret_cast<LMTrace *>(dv + c * sizeof(LMState));
  }
  if (e == hipSuccess) {
    static_assert(sizeof(PoseState) % 8 == 0 && sizeof(LMState) % 8 == 0, "8-byte aligned sub-blocks");
    b->d_poses = reinterpret_cast<PoseState *>(b->d_lm_block);
    b->d_states = reinterpret_cast<LMState *>(b->d_lm_block + c * sizeof(PoseState));
    b->d_traces = reinterpret_cast<LMTrace *>(b->d_lm_block + c * (sizeof(PoseState) + sizeof(LMState)));
    b->h_poses = reinterpret_cast<PoseState *>(b->h_lm_block);
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&b->dv_h_poses), b->h_poses, 0) != hipSuccess) { b->dv_h_poses = nullptr; (void)hipGetLastError(); }
    b->h_states = reinterpret_cast<LMState *>(b->h_lm_block + c * sizeof(PoseState));
    b->h_traces = reinterpret_cast<LMTrace *>(b->h_lm_block + c * (sizeof(PoseState) + sizeof(LMState)));
  }
  if (e == hipSuccess) {  // traces and states are read back row by row: never hand out uninitialised memory
    // (on the batch's own stream: a null-stream memset is not ordered with this non-blocking stream and could land
    // after the first solve's upload -- states wiped to "not running", the host waiting for a flag nobody lowers)
    e = hipMemsetAsync(b->d_lm_block, 0, lm_bytes, b->stream);
    std::memset(b->h_lm_block, 0, lm_bytes);
    std::memset(b->h_deliver, 0, c * (sizeof(LMState) + sizeof(LMTrace)));
  }
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(b->stream);
    batch_free_device(b);
    cached_stream_destroy(b->stream, b->device);
    delete b;
    return fail(EA_ERR_ALLOC, std::string("batch allocation: ") + hipGetErrorString(e));
  }
  *out = b;
  return EA_OK;
}

extern "C" void ea_batch_destroy(ea_batch *b) {
  if (!b) return;
  for (ea_batch *c : b->parts) ea_batch_destroy(c);
  b->parts.clear();
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  batch_free_device(b);
  if (b->own_stream && b->stream) cached_stream_destroy(b->stream, b->device);  // (synchronised above)
  for (ea_problem *p : b->probs)
    if (p && p->self == b) p->self = nullptr;
  delete b;
}

extern "C" int ea_batch_count(const ea_batch *b) { return b ? (int)b->probs.size() : 0; }
extern "C" void *ea_internal_batch_stream(ea_batch *b, int *device) {
  if (device) *device = b ? b->device : -1;
  return b ? (void *)b->stream : nullptr;
}

// (re)build descriptors and the tile list when any problem changed
static void fill_desc(const ea_problem *p, ProblemDesc &d) {
  std::memset(&d, 0, sizeof(d));
  d.x = p->d_x; d.y = p->d_y; d.z = p->d_z; d.dt = p->d_dt;
  d.dt32 = (p->dtype == EA_F64 && p->dt32_exact) ? p->d_dt32 : nullptr;
  d.n = (int32_t)p->n; d.W = p->W; d.H = p->H; d.pitch = p->pitch;
  d.fx = p->cam.fx; d.fy = p->cam.fy; d.cx = p->cam.cx; d.cy = p->cam.cy;
  d.loss_a = p->loss_a; d.z_guard = p->z_guard; d.z_eps = p->z_eps;
  d.fxf = (float)d.fx; d.fyf = (float)d.fy; d.cxf = (float)d.cx; d.cyf = (float)d.cy;
  d.loss_inv_b = 1.0 / (d.loss_a * d.loss_a); d.loss_inv_bf = (float)d.loss_inv_b;
  d.loss_af = (float)d.loss_a; d.z_guardf = (float)d.z_guard; d.z_epsf = (float)d.z_eps;
  d.loss_kind = p->loss_kind; d.rot_transposed = p->rot_transposed;
  d.variant = p->variant;
  for (int i = 0; i < 5; ++i) { d.dist[i] = p->dist[i]; d.distf[i] = (float)p->dist[i]; }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      d.A[3 * r + c] = p->T12[4 * r + c]; d.Ai[3 * r + c] = p->T12inv[4 * r + c];
      d.Af[3 * r + c] = (float)d.A[3 * r + c]; d.Aif[3 * r + c] = (float)d.Ai[3 * r + c];
    }
    d.d[r] = p->T12[4 * r + 3]; d.di[r] = p->T12inv[4 * r + 3];
    d.df[r] = (float)d.d[r]; d.dif[r] = (float)d.di[r];
  }
}

// (re)build descriptors when any problem (or one of its terms) changed
static int batch_build(ea_batch *b) {
  // every entry point that launches or allocates passes through here first: the calling thread's current device is the
  // batch's from now on (a process driving several GPUs may have switched between calls)
  HIPCHK(hipSetDevice(b->device));
  if (b->needs_drain) {  // launches of an abandoned solve may still be queued on (or stuck in) the stream
    HIPCHK(hipStreamSynchronize(b->stream));
    b->needs_drain = false;
  }
  uint64_t sig = 0;
  bool dirty = !b->built;
  for (size_t i = 0; i < b->probs.size(); ++i) {
    uint64_t v = b->probs[i]->version;
    for (ea_problem *tm : b->probs[i]->terms) v = v * 1000003u + tm->version + 17;
    if (b->versions[i] != v) dirty = true;
    sig += v;
  }
  (void)sig;
  if (!dirty) return EA_OK;
  b->built = false;  // until the last allocation and upload below has succeeded
  if (b->bench_graph) { (void)hipGraphExecDestroy(b->bench_graph); b->bench_graph = nullptr; b->bench_graph_steps = 0; b->bench_riding_steps = 0; }
  bench_ring_free(b);  // (sized by the row count of the old build)
  b->kp_K = 0;         // (resident pose states and the replicated tables were built for the old descriptors)
  b->kp_G = 0;
  // terms of a problem follow it; all share its pose
  std::vector<const ea_problem *> terms;
  std::vector<int> term_group;
  int64_t total = 0, max_n = 0;
  int any_variant = 0;
  for (size_t i = 0; i < b->probs.size(); ++i) {
    std::vector<const ea_problem *> fam;
    fam.push_back(b->probs[i]);
    for (ea_problem *tm : b->probs[i]->terms) fam.push_back(tm);
    for (const ea_problem *p : fam) {
      if (p->device != b->device || p->dtype != b->dtype) return fail(EA_ERR_INVALID_ARG, "terms must share device and dtype");
      if (!p->d_dt) return fail(EA_ERR_STATE, "distance-transform image not set (ea_problem_set_dt)");
      if (p->n > 0 && !p->d_x) return fail(EA_ERR_STATE, "edge points not set (ea_problem_set_points)");
      if (p->rot_transposed != b->probs[i]->rot_transposed) return fail(EA_ERR_INVALID_ARG, "terms of one problem must agree on rot_transposed");
      total += p->n;
      max_n = std::max<int64_t>(max_n, p->n);
      any_variant |= p->variant;
      terms.push_back(p);
      term_group.push_back((int)i);
    }
  }
  any_variant |= b->t_variant;  // a part of a larger batch runs the kernel form the whole batch runs
  b->any_variant = any_variant;
  b->terms_are_groups = terms.size() == b->probs.size() ? 1 : 0;
  // Launch shape, measured on MI355X (profiles/r01_sweep*.txt, r02_single_shape_sweep.txt, r02_batch_shape_sweep.txt): what
  // counts is the (evaluate, fold) step and the LM iteration, i.e. the kernel AND the number of partial rows behind it.
  // Small problems are latency-bound: one point per lane and as many waves as possible.  From ~1e5 points two points
  // per lane cost the kernel nothing and halve the rows; larger clouds take 1024-thread workgroups (fp32: 2-4 points per
  // lane on top): the kernel time hardly moves, the rows drop to a few hundred.
  int nt_auto = 256, ppt_auto = 1;
  if (terms.size() == 1) {
    if (b->dtype == EA_F32) {
      if (max_n >= 800000) { nt_auto = 1024; ppt_auto = 4; }
      else if (max_n >= 300000) { nt_auto = 1024; ppt_auto = 2; }
      else if (max_n >= 150000) { nt_auto = 1024; ppt_auto = 1; }   // 2e5: step 6.3 us, against 6.7 at 256 x 4
      else if (max_n >= 80000) ppt_auto = 2;                        // 1e5: step 5.7 us / solve 159 us, against 6.3 / 177
    } else {
      // fp64 runs ONE 1024-thread workgroup per CU (128-VGPR budget): that shape pays when its workgroups fill whole
      // rounds of the 256 CUs (3.4e5 points = 1.33 rounds: 8.8 us against 6.0 us at 256 x 2)
      const int64_t wg1024 = (max_n + 1023) / 1024, rounds = (wg1024 + 255) / 256;
      if (max_n >= 150000 && (rounds == 1 || wg1024 * 4 >= rounds * 256 * 3)) nt_auto = 1024;
      else if (max_n >= 80000) ppt_auto = 2;
    }
  } else if (b->dtype == EA_F32) {
    if (max_n >= 800000) { nt_auto = 1024; ppt_auto = 4; }
    else if (max_n >= 300000) { nt_auto = 1024; ppt_auto = 2; }
    else if (max_n >= 150000) { nt_auto = 256; ppt_auto = 4; }
    else {
      // batches of small problems: one point per lane (latency-bound up to ~32 pairs; in raster order two points per lane
      // cost 4-6 % at every size) -- except large batches stored tile by tile, where two points per lane halve the
      // butterfly per point and still read neighbouring texels: 64 x 50k 23.5 -> 21.4 us, 128 x 50k 47.7 -> 41.3 us
      // (0.63 -> 0.69 / 0.71 of the HBM roofline; profiles/r02_batch_shape_sweep.txt)
      bool all_tiled = true;
      for (const ea_problem *p : terms) all_tiled = all_tiled && p->order_tile_used > 0;
      ppt_auto = (all_tiled && total >= 2400000) ? 2 : 1;
    }
  } else {
    ppt_auto = total >= 80000 ? 2 : 1;  // same kernel time at 1e5 points, half the rows for the LM step to fold
  }
  if (terms.size() == 1 && b->t_fused != 0 && !any_variant) {
    // A single problem whose evaluation fits one workgroup per CU can run one launch per LM iteration (ea_lm_iter_kernel:
    // 256-thread workgroups, at most 255 chunks + the writer), which beats every pair-form shape measured
    // (profiles/r03_ab_fused_shapes.txt: fp64 7e4 points 13.1 -> 11.9 us per iteration, fp32 2e5 points 12.6 -> 12.0): take
    // the fewest points per lane that fit.
    const int max_ppt = b->dtype == EA_F32 ? 4 : 2;
    for (int pp = 1; pp <= max_ppt; pp *= 2)
      if ((max_n + 256 * pp - 1) / (256 * pp) <= 255) { nt_auto = 256; ppt_auto = pp; break; }
  }
  int nt = (b->t_nt == 1024 || b->t_nt == 256) ? b->t_nt : nt_auto;
  int ppt = b->t_ppt;
  if (ppt != 1 && ppt != 2 && ppt != 4) ppt = ppt_auto;
  if (nt == 1024 && b->dtype == EA_F64) ppt = 1;  // 128-VGPR budget at 16 waves/CU
  if (b->dtype == EA_F64 && ppt > 2) ppt = 2;
  if (any_variant) { nt = 256; ppt = std::min(ppt, 2); }  // the variant kernel is built for this shape only
  // raw-buffer addressing (32-bit byte offsets into the image): on unless an image reaches 2 GiB
  int buffer_loads = b->t_buf >= 0 ? (b->t_buf ? 1 : 0) : 1;
  for (const ea_problem *p : terms)
    if ((size_t)p->pitch * (size_t)(p->H + 2 * kImagePad) * (b->dtype == EA_F32 ? 4 : 8) >= ((size_t)1 << 31)) buffer_loads = 0;
  b->buffer_loads = buffer_loads;
  b->ppt = ppt;
  b->nt = nt;
  const int64_t chunk = (int64_t)nt * ppt;
  b->chunk = (int)chunk;
  std::vector<ProblemDesc> descs(terms.size());
  std::vector<GroupDesc> groups(b->probs.size());
  int rows = 0, max_chunks = 0;
  int64_t row_begin = 0;
  std::vector<int64_t> row_offsets(b->probs.size() + 1, 0);
  for (size_t k = 0; k < terms.size(); ++k) {
    const ea_problem *p = terms[k];
    ProblemDesc &d = descs[k];
    fill_desc(p, d);
    d.group = term_group[k];
    d.row_begin = row_begin;
    if (k == 0 || term_group[k - 1] != term_group[k]) row_offsets[(size_t)term_group[k]] = row_begin;
    row_begin += p->n;
    const int nchunks = (int)((p->n + chunk - 1) / chunk);
    d.tile_begin = rows;
    rows += nchunks;
    d.tile_end = rows;
    max_chunks = std::max(max_chunks, nchunks);
    GroupDesc &g = groups[term_group[k]];
    if (k == 0 || term_group[k - 1] != term_group[k]) { g.tile_begin = d.tile_begin; g.term_begin = (int)k; }
    g.tile_end = d.tile_end;
    g.term_end = (int)k + 1;
  }
  b->nterms = (int)terms.size();
  b->group0 = groups.empty() ? GroupDesc{0, 0, 0, 0} : groups[0];
  if (!descs.empty()) { b->x0 = descs[0].x; b->y0 = descs[0].y; b->z0 = descs[0].z; b->n0 = descs[0].n; }
  else { b->x0 = b->y0 = b->z0 = nullptr; b->n0 = 0; }
  b->ntiles = rows;
  b->max_chunks = max_chunks;
  row_offsets[b->probs.size()] = row_begin;
  b->row_offsets = row_offsets;
  b->total_rows = row_begin;
  b->max_n = max_n;
  if (b->nterms > b->terms_cap || !b->d_desc_block) {
    cached_free(b->d_desc_block);
    b->d_desc_block = nullptr; b->d_probs = nullptr; b->d_groups = nullptr;
    b->terms_cap = b->nterms + 8;
    HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_desc_block),
                         (size_t)b->terms_cap * sizeof(ProblemDesc) + b->probs.size() * sizeof(GroupDesc), b->device));
  }
  static_assert(sizeof(ProblemDesc) % alignof(GroupDesc) == 0, "the groups sit right behind the terms");
  b->d_probs = reinterpret_cast<ProblemDesc *>(b->d_desc_block);
  b->d_groups = reinterpret_cast<GroupDesc *>(b->d_desc_block + (size_t)b->nterms * sizeof(ProblemDesc));
  if (b->t_test_fail_build) {  // (tests/test_gpu_robustness.py: a build that fails here must leave the batch dirty)
    b->t_test_fail_build = 0;
    return fail(EA_ERR_ALLOC, "batch build: injected allocation failure (test hook)");
  }
  if (b->ntiles > b->tiles_cap) {
    cached_free(b->d_partials);
    b->d_partials = nullptr;
    b->tiles_cap = b->ntiles + b->ntiles / 4 + 16;
    HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_partials), (size_t)b->tiles_cap * kAccSlots * sizeof(double), b->device));
    // (stream-ordered on the batch's own stream: a null-stream memset is not ordered with a non-blocking stream and may
    // land after the first evaluation has written its rows)
    HIPCHK(hipMemsetAsync(b->d_partials, 0, (size_t)b->tiles_cap * kAccSlots * sizeof(double), b->stream));
  }
  {
    const size_t pb = descs.size() * sizeof(ProblemDesc), gb = groups.size() * sizeof(GroupDesc);
    if (!b->desc_done) HIPCHK(hipEventCreateWithFlags(&b->desc_done, hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(b->desc_done));  // the staging block is free again (it always is by now)
    if (b->h_desc_cap < pb + gb) {
      if (b->h_desc) cached_host_free(b->h_desc);
      b->h_desc = nullptr;
      b->h_desc_cap = ((pb + gb) * 2 + 1024 + 4095) & ~(size_t)4095;  // (whole pages: batches of equal shape find each other's block)
      HIPCHK(cached_host_malloc(reinterpret_cast<void **>(&b->h_desc), b->h_desc_cap, hipHostMallocDefault, b->device));
    }
    std::memcpy(b->h_desc, descs.data(), pb);
    std::memcpy(b->h_desc + pb, groups.data(), gb);
    HIPCHK(hipMemcpyAsync(b->d_desc_block, b->h_desc, pb + gb, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipEventRecord(b->desc_done, b->stream));
  }
  // LDS staging of the DT footprint is available but off by default: on MI355X the unaligned 16-byte
  // row loads served by the XCD's L2 beat it at every size measured (profiles/LOG.md section 5)
  int use_lds = b->t_use_lds < 0 ? 0 : b->t_use_lds;
  int lds = b->t_lds_bytes >= 0 ? b->t_lds_bytes : (b->dtype == EA_F32 ? 32768 : 49152);
  if (lds > 61440) lds = 61440;
  b->lds_bytes = (use_lds && !any_variant) ? lds : 0;
  b->wide = (b->t_wide > 0 && b->dtype == EA_F32 && !any_variant && b->lds_bytes == 0) ? 1 : 0;
  {
    bool all32 = b->dtype == EA_F64 && !any_variant && b->lds_bytes == 0 && b->t_img32 != 0 && !terms.empty();
    for (const ea_problem *p : terms) all32 = all32 && p->dt32_exact && p->d_dt32;
    b->img32 = all32 ? 1 : 0;
  }
  b->xcd_remap = b->t_xcd < 0 ? 1 : (b->t_xcd ? 1 : 0);
  // only now is the batch consistent with its problems: a failure above leaves it dirty, so the next call rebuilds
  // instead of launching on freed or missing buffers
  for (size_t i = 0; i < b->probs.size(); ++i) {
    uint64_t v = b->probs[i]->version;
    for (ea_problem *tm : b->probs[i]->terms) v = v * 1000003u + tm->version + 17;
    b->versions[i] = v;
  }
  b->built = true;
  return EA_OK;
}

static void host_pose_state(const ea_problem *p, const double *q, const double *t, PoseState *ps) {
  double x[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
  make_pose_state(x, p->rot_transposed, 1, ps);
}

static int batch_launch_eval(ea_batch *b, const PoseState *poses = nullptr) {
  HIPCHK(launch_eval_fused(b->dtype, b->ppt, b->nt, b->any_variant, b->d_probs, b->nterms, b->chunk, b->max_chunks,
                           b->xcd_remap, poses ? poses : b->d_poses, b->d_partials, b->lds_bytes, b->wide, b->terms_are_groups, b->buffer_loads, b->img32,
                           b->x0, b->y0, b->z0, b->n0, b->stream));
  return EA_OK;
}

static int batch_upload_poses(ea_batch *b, const double *q, const double *t) {
  const int count = (int)b->probs.size();
  for (int i = 0; i < count; ++i) host_pose_state(b->probs[i], q + 4 * i, t + 3 * i, &b->h_poses[i]);
  HIPCHK(hipMemcpyAsync(b->d_poses, b->h_poses, count * sizeof(PoseState), hipMemcpyHostToDevice, b->stream));
  b->poses_uploaded = true;
  b->poses_staged = false;
  return EA_OK;
}

// the measurement hooks re-evaluate at "the poses of the last evaluation": bring them to the device if that evaluation read them
// from host memory
static int ensure_poses_on_device(ea_batch *b) {
  if (b->poses_uploaded) return EA_OK;
  if (!b->poses_staged) return fail(EA_ERR_STATE, "no poses uploaded yet (ea_batch_bench_eval or ea_batch_eval first)");
  HIPCHK(hipMemcpyAsync(b->d_poses, b->h_poses, b->probs.size() * sizeof(PoseState), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->poses_uploaded = true;
  b->poses_staged = false;
  return EA_OK;
}

// the 32 accumulator slots of every problem (pinned host copy) -> the caller's cost / 6x6 JtJ / Jtr / invalid count
static void unpack_eval_out(const EvalOut *src, int count, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid) {
  for (int i = 0; i < count; ++i) {
    const double *acc = src[i].acc;
    if (cost) cost[i] = acc[kAccCost];
    if (JtJ) {
      int k = 0;
      for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) {
          JtJ[36 * i + 6 * a + c] = acc[kAccJtJ + k];
          JtJ[36 * i + 6 * c + a] = acc[kAccJtJ + k];
          ++k;
        }
    }
    if (Jtr) for (int a = 0; a < 6; ++a) Jtr[6 * i + a] = acc[kAccJtr + a];
    if (n_invalid) n_invalid[i] = (int64_t)llround(acc[kAccInvalid]);
  }
}
static void unpack_eval_out(const ea_batch *b, int count, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid) {
  unpack_eval_out(b->h_out, count, cost, JtJ, Jtr, n_invalid);
}

// The last fold of a synchronous evaluation raises a flag in pinned host memory once every result has landed there
// (ea_reduce_done_kernel); the host polls it instead of waiting for the stream's completion signal, which arrives ~10 us
// later.  The poll is bounded: should the flag not show up (it always has), the stream is synchronised the classic way --
// the results are complete then, and a device fault surfaces as the error it is.
static hipError_t launch_last_fold(ea_batch *b, const GroupDesc *groups, int count, const double *rows, EvalOut *out) {
  const size_t c = b->probs.size();
  b->done_seq = b->done_seq == 0x7fffffff ? 1 : b->done_seq + 1;
  return launch_reduce_done(groups, count, rows, out, b->d_done_count, b->d_progress + 3 * c, b->done_seq, b->stream);
}

static int wait_results(ea_batch *b) {
  const size_t c = b->probs.size();
  if (!b->t_poll) {
    HIPCHK(hipStreamSynchronize(b->stream));
    return EA_OK;
  }
  SpinWait wait(2000.0);
  while (__atomic_load_n(&b->h_progress[3 * c], __ATOMIC_ACQUIRE) != b->done_seq) {
    if (wait.poll()) {
      HIPCHK(hipStreamSynchronize(b->stream));
      return EA_OK;
    }
  }
  return EA_OK;
}

extern "C" int ea_batch_eval(ea_batch *b, const double *q, const double *t, double *cost, double *JtJ,
                             double *Jtr, int64_t *n_invalid) {
  if (!b || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  const int count = (int)b->probs.size();
  if (count <= 4 && b->dv_h_poses && b->t_zero_copy != 0) {
    // A lone evaluation of a few problems: the pose constants stay where the host wrote them (pinned memory the device can
    // read) and every workgroup fetches its ~100 bytes from there -- one PCIe round trip inside the kernel's head instead of a
    // DMA transfer in front of the launch (26 -> 24 us per call, profiles/r03_ab_zero_copy.txt).  The call is synchronous: nobody touches h_poses before the
    // results are back.  (d_poses keeps whatever an earlier call uploaded.)
    for (int i = 0; i < count; ++i) host_pose_state(b->probs[i], q + 4 * i, t + 3 * i, &b->h_poses[i]);
    b->poses_uploaded = false;
    b->poses_staged = true;
    rc = batch_launch_eval(b, b->dv_h_poses);
  } else {
    rc = batch_upload_poses(b, q, t);
    if (rc != EA_OK) return rc;
    rc = batch_launch_eval(b);
  }
  if (rc != EA_OK) return rc;
  // the fold writes its 256 bytes per problem straight into pinned host memory: no device-to-host copy behind it (-6 us
  // of a 31 us call); the kernel's end makes them visible
  HIPCHK(launch_last_fold(b, b->d_groups, count, b->d_partials, b->dv_out));
  if ((rc = wait_results(b)) != EA_OK) return rc;
  unpack_eval_out(b, count, cost, JtJ, Jtr, n_invalid);
  return EA_OK;
}

// ---- K evaluations at K different poses (ea_batch_set_poses / ea_batch_eval_resident_poses / ea_batch_eval_poses) ------
// What a caller that drives its own optimiser -- or probes a cost surface, or runs a line search -- asks of the evaluator:
// ceres::Problem::Evaluate once per pose (src/SolveEA.cpp:241 is the reference's one call of it).  One evaluation through
// ea_batch_eval is a launch pair and a synchronisation (~30 us on a 5e4-point pair, 3 us of which are the kernel, run by
// 196 workgroups on a 256-CU chip).  K independent evaluations do not have to queue up behind each other: the POSE is one
// more batch dimension.  The evaluation kernel already takes (workgroup column, term) grids with a descriptor and a pose
// per term; here the descriptor table is replicated G times -- copy g of term j keeps j's points and image and owns its own
// partial rows and pose slot g -- so ONE launch of (chunks, G x terms) workgroups evaluates every point at G poses, one
// fold launch of G x count workgroups folds them straight into pinned host memory (the sums of pose k are those
// ea_batch_eval returns at pose k up to rounding: the pose path cuts the points into chunks of its own, kposes_shape),
// and K poses are ceil(K / G) such pairs behind one synchronisation.  Every (point, pose) pair runs the whole per-point arithmetic; nothing is shared between poses but the
// bytes of the points and the image, which the later poses find in the caches.  (Round 3 first chained K launches with the
// fold of evaluation k-1 riding in launch k -- 3.3 us per C2 evaluation, one small launch at a time; G poses per launch
// fill the chip.)

// G: poses per launch -- enough workgroups to fill the chip several times over (~32k), within the grid's y limit and 64 MB
// of partial rows
// Launch shape.  One evaluation of a batch is shaped for latency (one point per lane on frame-sized problems: as many
// wavefronts as possible, few partial rows for the LM step to fold); a launch of G poses is throughput-bound, where more
// points per lane amortise the wavefront butterfly and 256-lane workgroups keep the occupancy
// (profiles/r03_ab_poses_shape.txt: C2 fp64 0.69 -> 0.58 us per evaluation at two points per lane, fp32 0.33 -> 0.27; C5
// fp32 5.65 -> 4.50 at 256 x 4 instead of 1024 x 4).  Explicit tuning ("points_per_thread", "threads") wins.
static void kposes_shape(ea_batch *b) {
  int nt = 256, ppt = b->dtype == EA_F64 ? 2 : (b->max_n >= 200000 ? 4 : 2);
  if (b->lds_bytes > 0 || b->wide) { nt = b->nt; ppt = b->ppt; }          // (those forms keep the shape they were tuned at)
  if (b->t_nt == 1024 || b->t_nt == 256) nt = b->t_nt;
  if (b->t_ppt == 1 || b->t_ppt == 2 || b->t_ppt == 4) ppt = b->t_ppt;
  if (nt == 1024 && b->dtype == EA_F64) ppt = 1;
  if (b->dtype == EA_F64 && ppt > 2) ppt = 2;
  if (b->any_variant) { nt = 256; ppt = std::min(ppt, 2); }
  b->kp_nt = nt; b->kp_ppt = ppt; b->kp_chunk = nt * ppt;
  const ProblemDesc *hd = reinterpret_cast<const ProblemDesc *>(b->h_desc);
  int rows = 0, widest = 0;
  for (int j = 0; j < b->nterms; ++j) {
    const int nchunks = (int)(((int64_t)hd[j].n + b->kp_chunk - 1) / b->kp_chunk);
    rows += nchunks; widest = std::max(widest, nchunks);
  }
  b->kp_ntiles = rows; b->kp_max_chunks = widest;
}

static int kposes_group(const ea_batch *b, int K) {
  const int64_t wgs = std::max<int64_t>(1, (int64_t)b->kp_ntiles);
  int64_t g = (32768 + wgs - 1) / wgs;
  g = std::min<int64_t>(g, 65535 / std::max(1, b->nterms));
  g = std::min<int64_t>(g, ((int64_t)64 << 20) / (wgs * kAccSlots * (int64_t)sizeof(double)));
  if (b->t_kp_G > 0) g = std::min<int64_t>(g, b->t_kp_G);
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, K));
}

static void kposes_free_tables(ea_batch *b) {
  cached_free(b->d_kprobs); cached_free(b->d_kgroups); cached_free(b->d_krows);
  b->d_kprobs = nullptr; b->d_kgroups = nullptr; b->d_krows = nullptr;
  b->kp_G = 0;
}

// the replicated descriptor / group tables and the partial rows of G poses; rebuilt when the batch was (kp_G = 0)
static int kposes_tables(ea_batch *b, int G) {
  if (b->kp_G >= G && !(b->t_kp_G > 0 && b->kp_G > b->t_kp_G)) return EA_OK;
  HIPCHK(hipStreamSynchronize(b->stream));
  kposes_free_tables(b);
  const size_t nterms = (size_t)b->nterms, count = b->probs.size(), rows = (size_t)b->kp_ntiles;
  const ProblemDesc *hd = reinterpret_cast<const ProblemDesc *>(b->h_desc);                          // (batch_build's staging block)
  const GroupDesc *hg = reinterpret_cast<const GroupDesc *>(b->h_desc + nterms * sizeof(ProblemDesc));
  // the partial rows of one pose in the pose path's own chunking (kposes_shape)
  std::vector<int32_t> row0(nterms + 1, 0);
  for (size_t j = 0; j < nterms; ++j) row0[j + 1] = row0[j] + (int32_t)(((int64_t)hd[j].n + b->kp_chunk - 1) / b->kp_chunk);
  std::vector<ProblemDesc> descs((size_t)G * nterms);
  std::vector<GroupDesc> groups((size_t)G * count);
  for (int g = 0; g < G; ++g) {
    for (size_t j = 0; j < nterms; ++j) {
      ProblemDesc d = hd[j];
      d.tile_begin = row0[j] + (int32_t)(g * rows); d.tile_end = row0[j + 1] + (int32_t)(g * rows);
      d.group += (int32_t)(g * count);
      descs[(size_t)g * nterms + j] = d;
    }
    for (size_t i = 0; i < count; ++i) {
      GroupDesc gd = hg[i];
      gd.tile_begin = row0[(size_t)hg[i].term_begin] + (int32_t)(g * rows); gd.tile_end = row0[(size_t)hg[i].term_end] + (int32_t)(g * rows);
      gd.term_begin += (int32_t)(g * nterms); gd.term_end += (int32_t)(g * nterms);
      groups[(size_t)g * count + i] = gd;
    }
  }
  HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_kprobs), std::max<size_t>(1, descs.size()) * sizeof(ProblemDesc), b->device));
  HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_kgroups), groups.size() * sizeof(GroupDesc), b->device));
  HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_krows), std::max<size_t>(1, (size_t)G * rows) * kAccSlots * sizeof(double), b->device));
  if (!descs.empty()) HIPCHK(hipMemcpyAsync(b->d_kprobs, descs.data(), descs.size() * sizeof(ProblemDesc), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_kgroups, groups.data(), groups.size() * sizeof(GroupDesc), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemsetAsync(b->d_krows, 0, std::max<size_t>(1, (size_t)G * rows) * kAccSlots * sizeof(double), b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));  // (the host vectors go out of scope)
  b->kp_G = G;
  return EA_OK;
}

static int kposes_reserve(ea_batch *b, int K) {
  const size_t count = b->probs.size();
  if (K <= b->kp_cap) return EA_OK;
  HIPCHK(hipStreamSynchronize(b->stream));  // (launches still reading the old arrays)
  kposes_free(b);
  const int cap = std::max(K, 8);
  const size_t n = (size_t)cap * count;
  HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_kqt), n * 7 * sizeof(double), b->device));
  HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_kposes), n * sizeof(PoseState), b->device));
  HIPCHK(cached_host_malloc(reinterpret_cast<void **>(&b->h_kqt), n * 7 * sizeof(double), hipHostMallocDefault, b->device));
  HIPCHK(cached_host_malloc(reinterpret_cast<void **>(&b->h_kout), n * sizeof(EvalOut), hipHostMallocMapped, b->device));
  HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->dv_kout), b->h_kout, 0));
  b->kp_cap = cap;
  return EA_OK;
}

extern "C" int ea_batch_set_poses(ea_batch *b, int K, const double *q, const double *t) {
  if (!b || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (K < 1 || K > (1 << 20)) return fail(EA_ERR_INVALID_ARG, "K out of range");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  b->kp_K = 0;
  if ((rc = kposes_reserve(b, K)) != EA_OK) return rc;
  if (b->kp_G == 0) kposes_shape(b);   // (once per build of the batch)
  if ((rc = kposes_tables(b, kposes_group(b, K))) != EA_OK) return rc;
  const size_t n = (size_t)K * b->probs.size();
  // (the previous upload out of the same staging block has been consumed: every call below ends with its pose kernel
  // enqueued behind the copy, and the evaluations that follow are synchronised before they return)
  HIPCHK(hipStreamSynchronize(b->stream));
  for (size_t i = 0; i < n; ++i) {
    double *d = b->h_kqt + 7 * i;
    d[0] = q[4 * i]; d[1] = q[4 * i + 1]; d[2] = q[4 * i + 2]; d[3] = q[4 * i + 3];
    d[4] = t[3 * i]; d[5] = t[3 * i + 1]; d[6] = t[3 * i + 2];
  }
  HIPCHK(hipMemcpyAsync(b->d_kqt, b->h_kqt, n * 7 * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(launch_make_poses(b->d_kqt, (int)n, (int)b->probs.size(), b->d_probs, b->d_groups, b->d_kposes, b->stream));
  b->kp_K = K;
  return EA_OK;
}

// the K evaluations of the resident poses on the batch's stream, G poses per launch pair, results into dv_kout[k * count + i]
static int enqueue_resident_poses(ea_batch *b, int K, bool folds = true, bool flag_last = false) {
  const int count = (int)b->probs.size(), G = b->kp_G;
  for (int start = 0; start < K; start += G) {
    const int g = std::min(G, K - start);
    HIPCHK(launch_eval_poses(b->dtype, b->kp_ppt, b->kp_nt, b->any_variant, b->d_kprobs, g * b->nterms, b->kp_chunk, b->kp_max_chunks,
                             b->xcd_remap, b->d_kposes + (size_t)start * count, b->d_krows, b->lds_bytes, b->wide,
                             b->terms_are_groups, b->buffer_loads, b->img32, b->x0, b->y0, b->z0, b->n0, b->stream));
    if (!folds) continue;
    if (flag_last && start + g >= K) HIPCHK(launch_last_fold(b, b->d_kgroups, g * count, b->d_krows, b->dv_kout + (size_t)start * count));
    else HIPCHK(launch_reduce(b->d_kgroups, g * count, b->d_krows, b->dv_kout + (size_t)start * count, b->stream));
  }
  return EA_OK;
}

extern "C" int ea_batch_eval_resident_poses(ea_batch *b, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid) {
  if (!b) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  const int K = b->kp_K;
  if (K < 1 || b->kp_G < 1) return fail(EA_ERR_STATE, "no poses resident (ea_batch_set_poses first; a change of the batch's problems drops them)");
  const int count = (int)b->probs.size();
  if ((rc = enqueue_resident_poses(b, K, true, true)) != EA_OK) return rc;
  if ((rc = wait_results(b)) != EA_OK) return rc;
  const size_t n = (size_t)K * (size_t)count;
  if (cost || JtJ || Jtr || n_invalid) unpack_eval_out(b->h_kout, (int)n, cost, JtJ, Jtr, n_invalid);
  return EA_OK;
}

// (ea_hip_dev.h) `reps` runs of the resident poses' launches between one event pair on the batch's stream: milliseconds
// per run, the device's own view of the K evaluations; evaluations_only: without the fold launches (what bench.py divides
// by the number of evaluation launches for the dominant kernel's duration); launches (nullable) = evaluation launches per run
extern "C" int ea_batch_bench_resident_poses(ea_batch *b, int reps, int evaluations_only, double *ms_per_run, int *launches) {
  if (!b || !ms_per_run || reps < 1) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = ea_batch_eval_resident_poses(b, nullptr, nullptr, nullptr, nullptr);
  if (rc != EA_OK) return rc;
  EventPair evp;
  HIPCHK(hipEventCreate(&evp.e0));
  HIPCHK(hipEventCreate(&evp.e1));
  // (the stream is held while the runs are enqueued: the launches execute back to back from the queue)
  HIPCHK(hipLaunchHostFunc(b->stream, [](void *) { std::this_thread::sleep_for(std::chrono::milliseconds(2)); }, nullptr));
  HIPCHK(hipEventRecord(evp.e0, b->stream));
  for (int i = 0; i < reps; ++i) if ((rc = enqueue_resident_poses(b, b->kp_K, !evaluations_only)) != EA_OK) return rc;
  HIPCHK(hipEventRecord(evp.e1, b->stream));
  HIPCHK(hipEventSynchronize(evp.e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, evp.e0, evp.e1));
  *ms_per_run = (double)ms / reps;
  if (launches) *launches = (b->kp_K + b->kp_G - 1) / b->kp_G;
  return EA_OK;
}

// (ea_hip_dev.h) The floor of the launch mechanism the sequences above run on: a hipGraph of `nodes` kernel nodes, each an
// EMPTY kernel of grid x block threads, replayed between one event pair -- milliseconds per node.  What a launch of any
// size costs when it does nothing; a kernel's duration cannot go below it, so (algorithmic bytes / floor) bounds the
// roofline fraction a small launch can reach.
namespace ea { hipError_t launch_empty(int grid, int block, hipStream_t stream); }
extern "C" int ea_bench_graph_floor(int device, int nodes, int grid, int block, double *ms_per_node) {
  if (!ms_per_node || nodes < 1 || grid < 1 || block < 1 || block > 1024) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = check_device(device);
  if (rc != EA_OK) return rc;
  HIPCHK(hipSetDevice(device));
  hipStream_t st = nullptr;
  HIPCHK(cached_stream_create(&st, device));
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  EventPair evp;
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < nodes && e == hipSuccess; ++i) e = launch_empty(grid, block, st);
  const hipError_t ee = hipStreamEndCapture(st, &graph);
  if (e == hipSuccess) e = ee;
  if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (graph) (void)hipGraphDestroy(graph);
  double best = 1e30;
  if (e == hipSuccess) e = hipEventCreate(&evp.e0);
  if (e == hipSuccess) e = hipEventCreate(&evp.e1);
  for (int r = 0; r < 5 && e == hipSuccess; ++r) {  // (first replay uploads the graph; best of the rest)
    e = hipEventRecord(evp.e0, st);
    if (e == hipSuccess) e = hipGraphLaunch(exec, st);
    if (e == hipSuccess) e = hipEventRecord(evp.e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(evp.e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, evp.e0, evp.e1);
    if (r > 0 && e == hipSuccess) best = std::min(best, (double)ms / nodes);
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  (void)hipStreamSynchronize(st);
  cached_stream_destroy(st, device);
  if (e != hipSuccess) return fail(EA_ERR_HIP, std::string("graph floor: ") + hipGetErrorString(e));
  *ms_per_node = best;
  return EA_OK;
}

extern "C" int ea_batch_eval_poses(ea_batch *b, int K, const double *q, const double *t, double *cost, double *JtJ, double *Jtr,
                                   int64_t *n_invalid) {
  int rc = ea_batch_set_poses(b, K, q, t);
  if (rc != EA_OK) return rc;
  return ea_batch_eval_resident_poses(b, cost, JtJ, Jtr, n_invalid);
}

static void fill_summary(const LMState &s, const LMTrace &tr, int64_t npts, double ms, ea_summary *out) {
  std::memset(out, 0, sizeof(*out));
  out->termination = s.termination;
  out->why = s.why;
  out->num_iterations = s.iteration;
  out->num_successful_steps = s.num_successful;
  out->num_unsuccessful_steps = s.num_unsuccessful;
  const bool no_start = s.why == EA_WHY_INITIAL_EVAL_FAILED;  // no trace row exists: costs as Ceres leaves them (-1)
  out->initial_cost = no_start ? -1.0 : tr.it_cost[0];
  out->final_cost = no_start ? -1.0 : s.cost;
  out->num_point_evals = (int64_t)s.num_evals * npts;
  out->total_time_ms = ms;
  const int ni = no_start ? 0 : std::min(s.iteration + 1, (int)EA_MAX_TRACE);
  for (int i = 0; i < ni; ++i) {
    out->it_cost[i] = tr.it_cost[i];
    out->it_cost_change[i] = tr.it_cost_change[i];
    out->it_gradient_max_norm[i] = tr.it_gradient_max_norm[i];
    out->it_step_norm[i] = tr.it_step_norm[i];
    out->it_relative_decrease[i] = tr.it_relative_decrease[i];
    out->it_radius[i] = tr.it_radius[i];
    out->it_successful[i] = tr.it_successful[i];
  }
}

// ---- solve: the trust-region loop of every problem of a batch, on the device ---------------------------------------
//
// One run = one batch on its own stream.  The loop runs on the device: each (evaluate, LM step) pair reads the pose
// the previous step published.  The step kernel reports progress into pinned host memory; the host only keeps a few
// pairs queued ahead of it and stops enqueueing when every problem has terminated (pairs that are already queued
// find `active == 0` and return at once).
struct SolveRun {
  ea_batch *b = nullptr;
  int first = 0;  // index of the run's first problem in the caller's arrays
  int count = 0, enq = 0, budget = 0, ahead = 2;
  int seen = 0;   // evaluations the device had completed at the last look (progress = this moved, or a pair went out)
  unsigned spins = 0;
  bool done = false, fetch = false, moved = false;
  bool fused = false;  // one launch per iteration (ea_lm_iter_kernel) instead of (evaluate, step) pairs
};

static double resolve_timeout_ms(const ea_options &o) { return o.solve_timeout_ms == 0.0 ? 5000.0 : o.solve_timeout_ms; }

static int solve_start(SolveRun &r, const ea_options &o, const LMOptions &lo, const double *q, const double *t) {
  ea_batch *b = r.b;
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  const int count = r.count = (int)b->probs.size();
  for (int i = 0; i < count; ++i) {
    lm_init(&b->h_states[i], &lo, q + 4 * i, t + 3 * i, b->probs[i]->rot_transposed);
    host_pose_state(b->probs[i], q + 4 * i, t + 3 * i, &b->h_poses[i]);
    b->h_progress[i] = 1;          // running
    b->h_progress[count + i] = 0;  // evaluations whose step kernel has started
    b->h_progress[2 * count + i] = 0;  // step kernels complete (posted on request: the point-sharded solve)
  }
  if (b->t_test_stall_ms > 0)  // (tests/test_gpu_robustness.py: a device that shows no progress must trip the deadline)
    HIPCHK(hipLaunchHostFunc(b->stream, [](void *ms) { std::this_thread::sleep_for(std::chrono::milliseconds((intptr_t)ms)); },
                             (void *)(intptr_t)b->t_test_stall_ms));
  HIPCHK(hipMemcpyAsync(b->d_lm_block, b->h_lm_block, (size_t)count * (sizeof(PoseState) + sizeof(LMState)),
                        hipMemcpyHostToDevice, b->stream));
  b->poses_staged = false;  // (h_poses now holds the start poses, d_poses gets them with this copy)
  // One launch per iteration when every workgroup of the evaluation can run the LM step itself for free: one
  // plain residual family per problem in 256-thread workgroups on the L2 path, and the whole grid (chunks + the writer,
  // rounded to the XCDs) resident at once -- the kernel holds one workgroup per CU (ea_lm_iter_kernel).
  {
    const int gx = b->xcd_remap ? 8 * ((b->max_chunks + 1 + 7) / 8) : b->max_chunks + 1;
    r.fused = b->t_fused != 0 && b->nt == 256 && !b->any_variant && b->lds_bytes == 0 && !b->wide &&
              b->terms_are_groups && b->max_chunks > 0 && (int64_t)gx * count <= 256;
    if (r.fused) {
      if (b->iter_alt_count < count) {
        cached_free(b->d_iter_alt);
        b->d_iter_alt = nullptr; b->iter_alt_count = 0;
        HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_iter_alt), (size_t)count * (sizeof(LMState) + sizeof(LMCold)), b->device));
        b->iter_alt_count = count;
      }
      if (b->tiles_cap_alt < b->tiles_cap) {
        cached_free(b->d_partials_alt);
        b->d_partials_alt = nullptr; b->tiles_cap_alt = 0;
        HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_partials_alt), (size_t)b->tiles_cap * kAccSlots * sizeof(double), b->device));
        b->tiles_cap_alt = b->tiles_cap;
      }
    }
    b->last_fused = r.fused ? 1 : 0;
  }
  r.ahead = o.iterations_per_sync > 0 ? o.iterations_per_sync : 2;  // measured: 2 beats 1, 3, 4, 6 by 1-3 %
  r.budget = o.max_num_iterations + 2;  // every pair consumes at least one iteration
  r.enq = 0; r.spins = 0; r.seen = 0; r.done = false; r.fetch = false; r.moved = true;
  return EA_OK;
}

// one look at the run's progress words; enqueues the next (evaluate, step) pair when the device is less than
// `ahead` pairs ahead of the host
static int solve_pump(SolveRun &r, const LMOptions &lo) {
  ea_batch *b = r.b;
  const int count = r.count;
  bool any = false;
  int done = 0;
  for (int i = 0; i < count; ++i) {
    any = any || (__atomic_load_n(&b->h_progress[i], __ATOMIC_ACQUIRE) != 0);
    done = std::max(done, __atomic_load_n(&b->h_progress[count + i], __ATOMIC_ACQUIRE));
  }
  if (!any) { r.done = true; r.moved = true; return EA_OK; }
  if (done != r.seen) { r.seen = done; r.moved = true; }
  if (r.enq < r.budget && r.enq - done < r.ahead) {
    if (r.fused) {
      // launch j = enq + 1 steps on what launch j - 1 left in the buffers of parity (j - 1) and evaluates into those of parity j;
      // launch 0 is the plain evaluation at the start pose
      if (r.enq == 0) {
        int rc = batch_launch_eval(b);
        if (rc != EA_OK) return rc;
      }
      LMState *st[2] = {b->d_states, reinterpret_cast<LMState *>(b->d_iter_alt)};
      LMCold *cold[2] = {b->d_cold, reinterpret_cast<LMCold *>(b->d_iter_alt + (size_t)b->iter_alt_count * sizeof(LMState))};
      double *rows[2] = {b->d_partials, b->d_partials_alt};
      const int in = r.enq & 1, out = in ^ 1;
      HIPCHK(launch_lm_iter(b->dtype, b->ppt, b->d_probs, count, b->chunk, b->max_chunks, b->xcd_remap, b->d_poses, rows[in], rows[out],
                            b->buffer_loads, b->img32, b->x0, b->y0, b->z0, b->n0, b->d_groups, st[in], st[out], cold[in], cold[out],
                            b->d_traces, lo, b->d_progress, b->dv_states, b->dv_traces, b->group0, /*post_done=*/0, b->stream));
    } else {
      int rc = batch_launch_eval(b);
      if (rc != EA_OK) return rc;
      HIPCHK(launch_lm_step(b->d_groups, count, b->d_partials, b->d_poses, b->d_states, b->d_cold, b->d_traces, lo,
                            b->d_progress, b->dv_states, b->dv_traces, b->group0, 0, b->stream));
    }
    ++r.enq;
    r.spins = 0;
    r.moved = true;
  } else if (r.enq >= r.budget) {
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int i = 0; i < count; ++i)
      r.fetch = r.fetch || (__atomic_load_n(&b->h_progress[i], __ATOMIC_ACQUIRE) != 0);
    r.done = true;
  } else if ((++r.spins & 0x3fff) == 0) {
    // nothing to enqueue and no progress for a while: make sure the stream is still healthy
    hipError_t qe = hipStreamQuery(b->stream);
    if (qe != hipSuccess && qe != hipErrorNotReady) return fail(EA_ERR_HIP, hipGetErrorString(qe));
  }
  return EA_OK;
}

// The step kernel that ended a problem's solve has already delivered its final state and trace rows into pinned
// host memory, in front of the flag polled above: no copy, and no wait for the launches queued ahead (they find
// every problem finished and drain behind our back; the next use of the stream is ordered after them anyway).
// Only a solve cut short by the launch budget has to be fetched the classic way.
static int solve_collect(SolveRun &r, const ea_options &o, bool want_traces) {
  ea_batch *b = r.b;
  const int count = r.count;
  if (r.fetch) {
    HIPCHK(hipMemcpyAsync(b->h_states, b->d_states, (size_t)count * (sizeof(LMState) + sizeof(LMTrace)),
                          hipMemcpyDeviceToHost, b->stream));
    if (r.fused && (r.enq & 1))  // a solve cut short is still running: every launch wrote, the last one into the other buffer
      HIPCHK(hipMemcpyAsync(b->h_states, b->d_iter_alt, (size_t)count * sizeof(LMState), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (r.fused)  // a problem that ended earlier stopped alternating buffers: its final state is the delivered one
      for (int i = 0; i < count; ++i)
        if (__atomic_load_n(&b->h_progress[i], __ATOMIC_ACQUIRE) == 0) b->h_states[i] = b->hd_states[i];
    return EA_OK;
  }
  std::memcpy(b->h_states, b->hd_states, (size_t)count * sizeof(LMState));
  if (want_traces || o.minimizer_progress_to_stdout) {
    for (int i = 0; i < count; ++i) {
      // rows 0 .. iteration were delivered; the rest of the pinned block is stale
      const int ni = std::min(b->hd_states[i].iteration + 1, (int)kTrace);
      LMTrace &d = b->h_traces[i];
      const LMTrace &sr = b->hd_traces[i];
      std::memcpy(d.it_cost, sr.it_cost, ni * sizeof(double));
      std::memcpy(d.it_cost_change, sr.it_cost_change, ni * sizeof(double));
      std::memcpy(d.it_gradient_max_norm, sr.it_gradient_max_norm, ni * sizeof(double));
      std::memcpy(d.it_step_norm, sr.it_step_norm, ni * sizeof(double));
      std::memcpy(d.it_relative_decrease, sr.it_relative_decrease, ni * sizeof(double));
      std::memcpy(d.it_radius, sr.it_radius, ni * sizeof(double));
      std::memcpy(d.it_successful, sr.it_successful, ni * sizeof(int));
    }
  }
  return EA_OK;
}

static void solve_report(const SolveRun &r, const ea_options &o, double ms, double *q, double *t, ea_summary *summaries) {
  const ea_batch *b = r.b;
  for (int i = 0; i < r.count; ++i) {
    const LMState &s = b->h_states[i];
    for (int k = 0; k < 4; ++k) q[4 * i + k] = s.x[k];
    for (int k = 0; k < 3; ++k) t[3 * i + k] = s.x[4 + k];
    const LMTrace &tr = b->h_traces[i];
    int64_t npts = b->probs[i]->n;
    for (ea_problem *tm : b->probs[i]->terms) npts += tm->n;
    if (summaries) fill_summary(s, tr, npts, ms, &summaries[i]);
    if (o.minimizer_progress_to_stdout) {
      std::printf("problem %d\niter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n", r.first + i);
      const int ni = s.why == EA_WHY_INITIAL_EVAL_FAILED ? 0 : std::min(s.iteration + 1, (int)kTrace);
      for (int it = 0; it < ni; ++it)
        std::printf("%4d  %.6e  % .2e    %.2e   %.2e  % .2e  %.2e\n", it, tr.it_cost[it], tr.it_cost_change[it],
                    tr.it_gradient_max_norm[it], tr.it_step_norm[it], tr.it_relative_decrease[it], tr.it_radius[it]);
    }
  }
}

// Sub-batches for the concurrent solve: contiguous slices of the parent's problems, each with its own stream and
// buffers, created on first use and kept.
static int batch_parts(ea_batch *b, int parts) {
  if ((int)b->parts.size() == parts) return EA_OK;
  for (ea_batch *c : b->parts) ea_batch_destroy(c);
  b->parts.clear();
  const int count = (int)b->probs.size();
  for (int k = 0; k < parts; ++k) {
    const int lo = (int)((int64_t)count * k / parts), hi = (int)((int64_t)count * (k + 1) / parts);
    ea_batch *c = nullptr;
    int rc = ea_batch_create(&c, b->probs.data() + lo, hi - lo);
    if (rc != EA_OK) return rc;
    b->parts.push_back(c);
  }
  return EA_OK;
}

extern "C" int ea_batch_solve(ea_batch *b, const ea_options *opt_in, double *q, double *t, ea_summary *summaries) {
  if (!b || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  static_assert(EA_MAX_TRACE == kTrace, "trace length mismatch");
  const auto t0 = std::chrono::steady_clock::now();
  ea_options o;
  if (opt_in) o = *opt_in; else ea_default_options(&o);
  if (int vrc = check_options(o)) return vrc;
  LMOptions lo;
  lo.max_num_iterations = o.max_num_iterations;
  lo.function_tolerance = o.function_tolerance;
  lo.gradient_tolerance = o.gradient_tolerance;
  lo.parameter_tolerance = o.parameter_tolerance;
  lo.initial_trust_region_radius = o.initial_trust_region_radius;
  lo.max_trust_region_radius = o.max_trust_region_radius;
  lo.min_trust_region_radius = o.min_trust_region_radius;
  lo.min_relative_decrease = o.min_relative_decrease;
  lo.min_lm_diagonal = o.min_lm_diagonal;
  lo.max_lm_diagonal = o.max_lm_diagonal;
  lo.max_num_consecutive_invalid_steps = o.max_num_consecutive_invalid_steps;
  lo.jacobi_scaling = o.jacobi_scaling;
  lo.strategy = o.strategy;
  const int count = (int)b->probs.size();
  // Concurrent halves: a batch of many problems is solved as two sub-batches on two streams, pumped by this one
  // thread.  Inside a batch the evaluation (all CUs busy) and the LM step (one workgroup per problem, pure latency)
  // alternate; with two streams one half's step runs under the other half's evaluation
  // (32 x C2: 0.38 -> 0.32 ms fp32, 0.57 -> 0.46 ms fp64; scripts/archive/split_batch_probe.py).  Every problem's arithmetic
  // is what it is in the one-stream solve of the same batch (same launch shape, same chunks, same order of summation).
  int parts = b->t_streams > 0 ? b->t_streams : (count >= 16 ? 2 : 1);
  parts = std::max(1, std::min(parts, std::min(count, 8)));
  std::vector<SolveRun> runs((size_t)parts);
  if (parts == 1) {
    runs[0].b = b;
  } else {
    // the parts take the launch shape the whole batch resolves to (the heuristics look at the batch's totals), so that
    // every problem is cut into the same chunks, and summed in the same order, as in the one-stream solve
    int rc = batch_build(b);
    if (rc != EA_OK) return rc;
    rc = batch_parts(b, parts);
    if (rc != EA_OK) return rc;
    int first = 0;
    for (int k = 0; k < parts; ++k) {
      ea_batch *c = b->parts[(size_t)k];
      const int use_lds = b->lds_bytes > 0 ? 1 : 0;
      if (c->t_lds_bytes != b->t_lds_bytes || c->t_ppt != b->ppt || c->t_use_lds != use_lds || c->t_xcd != b->xcd_remap ||
          c->t_nt != b->nt || c->t_variant != b->any_variant || c->t_buf != b->buffer_loads || c->t_wide != b->t_wide ||
          c->t_img32 != (b->img32 ? -1 : 0)) {
        c->t_lds_bytes = b->t_lds_bytes; c->t_ppt = b->ppt; c->t_use_lds = use_lds; c->t_xcd = b->xcd_remap; c->t_nt = b->nt;
        c->t_variant = b->any_variant; c->t_buf = b->buffer_loads; c->t_wide = b->t_wide; c->t_img32 = b->img32 ? -1 : 0;
        c->built = false;
      }
      runs[(size_t)k].b = c;
      runs[(size_t)k].first = first;
      first += (int)c->probs.size();
    }
  }
  for (SolveRun &r : runs) {
    int rc = solve_start(r, o, lo, q + 4 * r.first, t + 3 * r.first);
    if (rc != EA_OK) return rc;
  }
  // The host polls pinned progress words; a deadline bounds the wait: if the device shows no progress (no evaluation
  // completed, nothing to enqueue) for solve_timeout_ms, give up with an error instead of spinning for ever on a kernel
  // that never lowers its flag.  The batches are marked for a drain before their next use.
  SpinWait wait(resolve_timeout_ms(o));
  for (;;) {
    bool all_done = true, moved = false;
    for (SolveRun &r : runs) {
      if (r.done) continue;
      r.moved = false;
      int rc = solve_pump(r, lo);
      if (rc != EA_OK) return rc;
      moved = moved || r.moved;
      all_done = all_done && r.done;
    }
    if (all_done) break;
    if (moved) wait.progress();
    else if (wait.poll()) {
      for (SolveRun &r : runs) r.b->needs_drain = true;
      b->needs_drain = true;
      char msg[160];
      std::snprintf(msg, sizeof msg, "solve deadline: no progress on the device for %.0f ms (ea_options.solve_timeout_ms)", wait.timeout_ms());
      return fail(EA_ERR_HIP, msg);
    }
  }
  for (SolveRun &r : runs) {
    int rc = solve_collect(r, o, summaries != nullptr);
    if (rc != EA_OK) return rc;
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (SolveRun &r : runs) solve_report(r, o, ms, q + 4 * r.first, t + 3 * r.first, summaries ? summaries + r.first : nullptr);
  return EA_OK;
}

extern "C" int ea_batch_bench_eval(ea_batch *b, const double *q, const double *t, int warmup, int steps,
                                   double *ms_total, double *ms_eval_kernel) {
  if (!b || !q || !t || steps < 1 || warmup < 0) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  const int count = (int)b->probs.size();
  rc = batch_upload_poses(b, q, t);
  if (rc != EA_OK) return rc;
  EventPair evp;
  HIPCHK(hipEventCreate(&evp.e0));
  HIPCHK(hipEventCreate(&evp.e1));
  const hipEvent_t e0 = evp.e0, e1 = evp.e1;
  auto one_step = [&]() -> int {
    int r = batch_launch_eval(b);
    if (r != EA_OK) return r;
    HIPCHK(launch_reduce(b->d_groups, count, b->d_partials, b->d_out, b->stream));
    return EA_OK;
  };
  for (int i = 0; i < warmup; ++i) if ((rc = one_step()) != EA_OK) return rc;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipEventRecord(e0, b->stream));
  for (int i = 0; i < steps; ++i) if ((rc = one_step()) != EA_OK) return rc;
  HIPCHK(hipEventRecord(e1, b->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  if (ms_total) *ms_total = ms;
  if (ms_eval_kernel) {
    // second pass over the same steps: an event pair around every launch of the per-point kernel
    const int m = std::min(steps, 64);
    EventList evl;
    evl.ev.assign(2 * (size_t)m, nullptr);
    std::vector<hipEvent_t> &ev = evl.ev;
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (int i = 0; i < m; ++i) {
      HIPCHK(hipEventRecord(ev[2 * i], b->stream));
      if ((rc = batch_launch_eval(b)) != EA_OK) return rc;
      HIPCHK(hipEventRecord(ev[2 * i + 1], b->stream));
      HIPCHK(launch_reduce(b->d_groups, count, b->d_partials, b->d_out, b->stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    double sum = 0.0;
    for (int i = 0; i < m; ++i) {
      float k = 0.f;
      HIPCHK(hipEventElapsedTime(&k, ev[2 * i], ev[2 * i + 1]));
      sum += k;
    }
    *ms_eval_kernel = sum / m;
  }
  return EA_OK;
}

// Capture `steps` x (evaluation + fold) into a hipGraph (untimed set-up).  A step is two dependent launches of ~3 us
// each; enqueued one by one the host needs ~3.1 us per launch and the stream never runs ahead of it
// (profiles/r02_bench_bracket.txt) -- replayed from a graph the same launches execute back to back from the queue.
extern "C" int ea_batch_bench_capture(ea_batch *b, int steps) {
  if (!b || steps < 1) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if (int prc = ensure_poses_on_device(b)) return prc;
  if (b->bench_graph) { (void)hipGraphExecDestroy(b->bench_graph); b->bench_graph = nullptr; b->bench_graph_steps = 0; b->bench_riding_steps = 0; }
  const int count = (int)b->probs.size();
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
  hipError_t e = hipSuccess;
  for (int i = 0; i < steps && e == hipSuccess; ++i) {
    if (batch_launch_eval(b) != EA_OK) { e = hipErrorUnknown; break; }
    e = launch_reduce(b->d_groups, count, b->d_partials, b->d_out, b->stream);
  }
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(b->stream, &graph);
  if (e != hipSuccess || ee != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    return fail(EA_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(e != hipSuccess ? e : ee));
  }
  e = hipGraphInstantiate(&b->bench_graph, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { b->bench_graph = nullptr; return fail(EA_ERR_HIP, std::string("graph instantiate: ") + hipGetErrorString(e)); }
  b->bench_graph_steps = steps;
  HIPCHK(hipGraphLaunch(b->bench_graph, b->stream));  // one untimed replay: uploads the executable graph
  HIPCHK(hipStreamSynchronize(b->stream));
  return EA_OK;
}

static int bench_ring_ensure(ea_batch *b) {
  const int count = (int)b->probs.size();
  const size_t row_doubles = (size_t)b->tiles_cap * kAccSlots;
  if (b->bench_ring == 2) return EA_OK;
  HIPCHK(hipStreamSynchronize(b->stream));
  bench_ring_free(b);
  HIPCHK(hipMalloc(&b->d_bench_rows, row_doubles * sizeof(double) * 2));
  hipError_t ea = hipMalloc(&b->d_bench_out, sizeof(EvalOut) * (size_t)count * 2);
  if (ea != hipSuccess) { bench_ring_free(b); return fail(EA_ERR_ALLOC, "bench row buffers: allocation failed"); }
  b->bench_ring = 2;
  HIPCHK(hipMemsetAsync(b->d_bench_rows, 0, row_doubles * sizeof(double) * 2, b->stream));
  HIPCHK(hipMemsetAsync(b->d_bench_out, 0, sizeof(EvalOut) * (size_t)count * 2, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return EA_OK;
}

// K steps as K launches + 1 on the batch's stream: evaluation k with the fold of step k-1 riding in it, rows alternating
// between the two arrays of the ring, a stand-alone fold of the last step's rows into the batch's result array.
static hipError_t enqueue_riding_steps(ea_batch *b, int steps) {
  const int count = (int)b->probs.size();
  const size_t row_doubles = (size_t)b->tiles_cap * kAccSlots;
  double *const own_rows = b->d_partials;
  hipError_t e = hipSuccess;
  for (int i = 0; i < steps && e == hipSuccess; ++i) {
    double *rows = b->d_bench_rows + row_doubles * (size_t)(i & 1);
    if (i == 0) {
      b->d_partials = rows;
      const int lr = batch_launch_eval(b);
      b->d_partials = own_rows;
      if (lr != EA_OK) e = hipErrorUnknown;
    } else {
      const int prev = (i - 1) & 1;
      e = launch_eval_fold(b->dtype, b->ppt, b->nt, b->d_probs, b->nterms, b->chunk, b->max_chunks, b->xcd_remap, b->d_poses,
                           rows, b->buffer_loads, b->img32, b->x0, b->y0, b->z0, b->n0, b->d_groups,
                           b->d_bench_rows + row_doubles * (size_t)prev, b->d_bench_out + (size_t)count * (size_t)prev, b->stream);
    }
  }
  if (e == hipSuccess)
    e = launch_reduce_nt(b->nt, b->d_groups, count, b->d_bench_rows + row_doubles * (size_t)((steps - 1) & 1), b->d_out, b->stream);
  return e;
}

// The same K steps with the fold off the evaluations' critical path.  A step is two dependent launches, and the second
// one (one workgroup per problem) leaves the chip idle behind a kernel boundary; the next evaluation does not need its
// result -- the K evaluations of the timed region are independent passes at the resident poses.  Here the fold of step
// k-1 RIDES in the launch of evaluation k (ea_eval_fold_kernel: one extra workgroup per problem), the evaluations
// alternating between two row arrays; a stand-alone fold closes the sequence.  K steps = K launches + 1, every step
// still runs its evaluation and its fold in full, the evaluation kernels still execute one after the other.  The folds
// sum in the order of a workgroup of the evaluation's size (equal to ea_batch_eval's 1024-thread fold up to rounding).
// (A two-branch graph with the folds on a side stream was measured first: the cross-branch edges of a replayed graph
// cost more than the fold -- 9-19 us per step against 5.5 serial, profiles/r02_ab_pipeline.txt.)
extern "C" int ea_batch_bench_capture_pipelined(ea_batch *b, int steps) {
  if (!b || steps < 1) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if (int prc = ensure_poses_on_device(b)) return prc;
  if (b->any_variant || !b->terms_are_groups || b->lds_bytes > 0 || b->wide)
    return fail(EA_ERR_STATE, "the pipelined form covers plain single-family problems on the L2 path");
  if (b->bench_graph) { (void)hipGraphExecDestroy(b->bench_graph); b->bench_graph = nullptr; b->bench_graph_steps = 0; b->bench_riding_steps = 0; }
  if ((rc = bench_ring_ensure(b)) != EA_OK) return rc;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
  hipError_t e = enqueue_riding_steps(b, steps);
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(b->stream, &graph);
  if (e != hipSuccess || ee != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    return fail(EA_ERR_HIP, std::string("graph capture (pipelined): ") + hipGetErrorString(e != hipSuccess ? e : ee));
  }
  e = hipGraphInstantiate(&b->bench_graph, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { b->bench_graph = nullptr; return fail(EA_ERR_HIP, std::string("graph instantiate: ") + hipGetErrorString(e)); }
  b->bench_graph_steps = steps;
  b->bench_riding_steps = steps;
  HIPCHK(hipGraphLaunch(b->bench_graph, b->stream));  // one untimed replay: uploads the executable graph
  HIPCHK(hipStreamSynchronize(b->stream));
  return EA_OK;
}

// The same K launches + 1 enqueued launch by launch instead of replayed from a graph: the first evaluation starts while
// the host is still enqueueing the others (a graph replay hands over all K + 1 packets before the first one runs), which
// is what a short sequence wants -- see profiles/r02_riding_eager_vs_graph.txt.  host_us as in ea_batch_bench_steps.
extern "C" int ea_batch_bench_steps_riding(ea_batch *b, int steps, double *host_us) {
  if (!b || steps < 1) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if (int prc = ensure_poses_on_device(b)) return prc;
  if (b->any_variant || !b->terms_are_groups || b->lds_bytes > 0 || b->wide)
    return fail(EA_ERR_STATE, "the pipelined form covers plain single-family problems on the L2 path");
  if ((rc = bench_ring_ensure(b)) != EA_OK) return rc;
  if (host_us && (!b->bench_e0 || !b->bench_e1)) {
    HIPCHK(hipEventCreate(&b->bench_e0));
    HIPCHK(hipEventCreate(&b->bench_e1));
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (host_us) HIPCHK(hipEventRecord(b->bench_e0, b->stream));
  const hipError_t e = enqueue_riding_steps(b, steps);
  if (e != hipSuccess) return fail(EA_ERR_HIP, std::string("riding steps: ") + hipGetErrorString(e));
  b->bench_riding_steps = steps;
  if (host_us) HIPCHK(hipEventRecord(b->bench_e1, b->stream));
  const auto t1 = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(b->stream));
  if (host_us) {
    host_us[0] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    host_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->bench_e0, b->bench_e1));
    host_us[2] = ms;
  }
  return EA_OK;
}

// The last-but-one step's result of a pipelined sequence (result slot (steps - 2) & 1): with ea_batch_bench_result this
// covers both a riding fold and the closing stand-alone fold.
extern "C" int ea_batch_bench_result_riding(ea_batch *b, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid) {
  if (!b) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!b->built || b->bench_ring != 2 || b->bench_riding_steps < 2)
    return fail(EA_ERR_STATE, "no pipelined sequence of at least two steps captured or run");
  const int count = (int)b->probs.size();
  HIPCHK(hipMemcpyAsync(b->h_out, b->d_bench_out + (size_t)count * (size_t)((b->bench_riding_steps - 2) & 1), count * sizeof(EvalOut),
                        hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  unpack_eval_out(b, count, cost, JtJ, Jtr, n_invalid);
  return EA_OK;
}

// What the last step of the last ea_batch_bench_steps left in the batch's result array (cost, JtJ, Jtr per problem, the
// layout of ea_batch_eval): lets a caller check that the timed launches computed what ea_batch_eval computes.
extern "C" int ea_batch_bench_result(ea_batch *b, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid) {
  if (!b) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!b->built || !(b->poses_uploaded || b->poses_staged)) return fail(EA_ERR_STATE, "nothing evaluated yet");
  const int count = (int)b->probs.size();
  HIPCHK(hipMemcpyAsync(b->h_out, b->d_out, count * sizeof(EvalOut), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  unpack_eval_out(b, count, cost, JtJ, Jtr, n_invalid);
  return EA_OK;
}

extern "C" int ea_batch_bench_steps(ea_batch *b, int steps, double *host_us) {
  if (!b || steps < 1) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if (int prc = ensure_poses_on_device(b)) return prc;
  const int count = (int)b->probs.size();
  // host_us != NULL: also an event pair around the region on the stream ([2] = milliseconds between them): the device's
  // own view of the K steps, from which bench.py takes the evaluation kernel's share of a step
  if (host_us && (!b->bench_e0 || !b->bench_e1)) {
    HIPCHK(hipEventCreate(&b->bench_e0));
    HIPCHK(hipEventCreate(&b->bench_e1));
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (host_us) HIPCHK(hipEventRecord(b->bench_e0, b->stream));
  if (b->bench_graph && b->bench_graph_steps == steps) {
    HIPCHK(hipGraphLaunch(b->bench_graph, b->stream));
  } else {
    for (int i = 0; i < steps; ++i) {
      if ((rc = batch_launch_eval(b)) != EA_OK) return rc;
      HIPCHK(launch_reduce(b->d_groups, count, b->d_partials, b->d_out, b->stream));
    }
  }
  if (host_us) HIPCHK(hipEventRecord(b->bench_e1, b->stream));
  const auto t1 = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(b->stream));
  if (host_us) {
    host_us[0] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    host_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->bench_e0, b->bench_e1));
    host_us[2] = ms;
  }
  return EA_OK;
}

// Average duration of the per-point kernel when `launches` of them are queued back to back between one event pair:
// the command processor dispatches launch i+1 while launch i executes, so the figure is the kernel's execution
// window (what rocprofv3 --kernel-trace reports), not execution + dispatch as an event pair around a lone launch.
extern "C" int ea_batch_bench_kernel(ea_batch *b, const double *q, const double *t, int warmup, int launches,
                                     double *ms_per_launch) {
  if (!b || !q || !t || !ms_per_launch || launches < 1 || warmup < 0) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  rc = batch_upload_poses(b, q, t);
  if (rc != EA_OK) return rc;
  EventPair evp;
  HIPCHK(hipEventCreate(&evp.e0));
  HIPCHK(hipEventCreate(&evp.e1));
  const hipEvent_t e0 = evp.e0, e1 = evp.e1;
  for (int i = 0; i < warmup; ++i) if ((rc = batch_launch_eval(b)) != EA_OK) return rc;
  HIPCHK(hipStreamSynchronize(b->stream));
  // hold the stream on a host function while the whole run is enqueued: the launches then execute from the queue,
  // back to back, however fast this thread happens to enqueue them
  HIPCHK(hipLaunchHostFunc(b->stream, [](void *) { std::this_thread::sleep_for(std::chrono::milliseconds(3)); }, nullptr));
  HIPCHK(hipEventRecord(e0, b->stream));
  for (int i = 0; i < launches; ++i) if ((rc = batch_launch_eval(b)) != EA_OK) return rc;
  HIPCHK(hipEventRecord(e1, b->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = (double)ms / launches;
  return EA_OK;
}

// the same measurement for the fold kernel of ea_batch_eval (ea_reduce_kernel over the rows the last evaluation left)
extern "C" int ea_batch_bench_fold(ea_batch *b, int warmup, int launches, double *ms_per_launch) {
  if (!b || !ms_per_launch || launches < 1 || warmup < 0) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  const int count = (int)b->probs.size();
  EventPair evp;
  HIPCHK(hipEventCreate(&evp.e0));
  HIPCHK(hipEventCreate(&evp.e1));
  const hipEvent_t e0 = evp.e0, e1 = evp.e1;
  for (int i = 0; i < warmup; ++i) HIPCHK(launch_reduce(b->d_groups, count, b->d_partials, b->d_out, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipLaunchHostFunc(b->stream, [](void *) { std::this_thread::sleep_for(std::chrono::milliseconds(3)); }, nullptr));
  HIPCHK(hipEventRecord(e0, b->stream));
  for (int i = 0; i < launches; ++i) HIPCHK(launch_reduce(b->d_groups, count, b->d_partials, b->d_out, b->stream));
  HIPCHK(hipEventRecord(e1, b->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = (double)ms / launches;
  return EA_OK;
}

// ---- materialised mode (the "EAResidue batch Evaluate" view) ------------------------------------------------
// Residual and 1x6 row of every point of every term, in the batch's dtype, into device arrays: the caller's or the
// library's own.  Rows of problem i are [offsets[i], offsets[i+1]) (ea_batch_row_offsets), terms of a problem adjacent,
// points in their storage order (ea_problem_get_points).

static int rows_own_buffers(ea_batch *b) {
  if (b->rows_cap >= b->total_rows && b->d_rows_r) return EA_OK;
  (void)hipFree(b->d_rows_r); (void)hipFree(b->d_rows_J);
  b->d_rows_r = b->d_rows_J = nullptr; b->rows_cap = 0;
  const size_t es = b->dtype == EA_F32 ? 4 : 8;
  const int64_t cap = std::max<int64_t>(b->total_rows, 1);
  HIPCHK(hipMalloc(&b->d_rows_r, (size_t)cap * es));
  hipError_t e = hipMalloc(&b->d_rows_J, (size_t)cap * 6 * es);
  if (e != hipSuccess) { (void)hipFree(b->d_rows_r); b->d_rows_r = nullptr; return fail(EA_ERR_ALLOC, "row arrays: allocation failed"); }
  b->rows_cap = cap;
  return EA_OK;
}

// a caller's device pointer: non-NULL, 16-byte aligned, and -- where this runtime knows the allocation -- device memory of
// the batch's GPU.  (Memory of another HIP runtime in the same process, e.g. the storage of a PyTorch-ROCm tensor, is
// unknown to hipPointerGetAttributes here although kernels can address it: such pointers are taken on trust.)
static int check_device_pointer(const ea_batch *b, const void *ptr, const char *what) {
  if (!ptr) return fail(EA_ERR_INVALID_ARG, std::string(what) + " is NULL");
  if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return fail(EA_ERR_INVALID_ARG, std::string(what) + " must be 16-byte aligned");
  hipPointerAttribute_t at;
  const hipError_t e = hipPointerGetAttributes(&at, ptr);
  if (e != hipSuccess) { (void)hipGetLastError(); return EA_OK; }
  if (at.type == hipMemoryTypeHost) return fail(EA_ERR_INVALID_ARG, std::string(what) + " is host memory (device memory expected)");
  if ((at.type == hipMemoryTypeDevice) && at.device != b->device)
    return fail(EA_ERR_INVALID_ARG, std::string(what) + " lives on another device than the batch");
  return EA_OK;
}

static int rows_launch(ea_batch *b, int corrected, int layout, int staged, int nontemporal, void *r_dev, void *J_dev) {
  HIPCHK(launch_eval_rows(b->dtype, b->any_variant, b->buffer_loads, b->img32, layout, staged, b->d_probs, b->nterms, b->max_n, b->d_poses,
                          corrected ? 1 : 0, nontemporal, b->total_rows, r_dev, J_dev, b->d_rows_invalid, b->stream));
  return EA_OK;
}

static int rows_prepare(ea_batch *b, const double *q, const double *t, int layout, void **r_dev, void **J_dev, int64_t capacity_rows) {
  if (!b || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (layout != 0 && layout != 1) return fail(EA_ERR_INVALID_ARG, "layout: 0 = J row-major [rows][6], 1 = column-major [6][rows]");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if ((*r_dev == nullptr) != (*J_dev == nullptr)) return fail(EA_ERR_INVALID_ARG, "r_dev and J_dev: both or neither");
  if (*r_dev) {
    if (capacity_rows < b->total_rows) return fail(EA_ERR_INVALID_ARG, "capacity_rows is smaller than the batch's row count (ea_batch_row_offsets)");
    if ((rc = check_device_pointer(b, *r_dev, "r_dev")) != EA_OK) return rc;
    if ((rc = check_device_pointer(b, *J_dev, "J_dev")) != EA_OK) return rc;
  } else {
    if ((rc = rows_own_buffers(b)) != EA_OK) return rc;
    *r_dev = b->d_rows_r; *J_dev = b->d_rows_J;
  }
  if (!b->d_rows_invalid) HIPCHK(hipMalloc(&b->d_rows_invalid, sizeof(unsigned int)));
  return batch_upload_poses(b, q, t);
}

extern "C" int ea_batch_row_offsets(ea_batch *b, int64_t *offsets) {
  if (!b || !offsets) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  for (size_t i = 0; i < b->row_offsets.size(); ++i) offsets[i] = b->row_offsets[i];
  return EA_OK;
}

extern "C" int ea_batch_eval_rows_device(ea_batch *b, const double *q, const double *t, int corrected, int layout, void *r_dev,
                                         void *J_dev, int64_t capacity_rows, int64_t *n_invalid) {
  int rc = rows_prepare(b, q, t, layout, &r_dev, &J_dev, capacity_rows);
  if (rc != EA_OK) return rc;
  HIPCHK(hipMemsetAsync(b->d_rows_invalid, 0, sizeof(unsigned int), b->stream));
  const int staged = b->t_rows_staged < 0 ? 1 : (b->t_rows_staged ? 1 : 0), nt = b->t_rows_nt < 0 ? 0 : (b->t_rows_nt ? 1 : 0);
  if ((rc = rows_launch(b, corrected, layout, staged, nt, r_dev, J_dev)) != EA_OK) return rc;
  unsigned int bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, b->d_rows_invalid, sizeof(bad), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));  // the rows are complete (and visible to any other stream) when this returns
  if (n_invalid) *n_invalid = (int64_t)bad;
  return EA_OK;
}

// the same into host arrays of the batch's dtype (library-owned device arrays + one copy each)
extern "C" int ea_batch_eval_rows(ea_batch *b, const double *q, const double *t, int corrected, int layout, void *r_host,
                                  void *J_host, int64_t capacity_rows, int64_t *n_invalid) {
  if (!b) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!r_host && !J_host) return fail(EA_ERR_INVALID_ARG, "r_host and J_host are both NULL");
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if (capacity_rows < b->total_rows) return fail(EA_ERR_INVALID_ARG, "capacity_rows is smaller than the batch's row count (ea_batch_row_offsets)");
  rc = ea_batch_eval_rows_device(b, q, t, corrected, layout, nullptr, nullptr, 0, n_invalid);
  if (rc != EA_OK) return rc;
  const size_t es = b->dtype == EA_F32 ? 4 : 8;
  if (b->total_rows > 0) {
    if (r_host) HIPCHK(hipMemcpy(r_host, b->d_rows_r, (size_t)b->total_rows * es, hipMemcpyDeviceToHost));
    if (J_host) HIPCHK(hipMemcpy(J_host, b->d_rows_J, (size_t)b->total_rows * 6 * es, hipMemcpyDeviceToHost));
  }
  return EA_OK;
}

// `launches` of the materialised-mode kernel queued back to back between one event pair (the stream held on a host
// function while they are enqueued), average execution window per launch.  mode: bit 0 = LDS-staged row-major stores,
// bit 1 = non-temporal stores.  r_dev / J_dev NULL: the library's own arrays.
extern "C" int ea_batch_bench_rows(ea_batch *b, const double *q, const double *t, int corrected, int layout, int mode, void *r_dev,
                                   void *J_dev, int64_t capacity_rows, int warmup, int launches, double *ms_per_launch) {
  if (!ms_per_launch || launches < 1 || warmup < 0) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = rows_prepare(b, q, t, layout, &r_dev, &J_dev, capacity_rows);
  if (rc != EA_OK) return rc;
  HIPCHK(hipMemsetAsync(b->d_rows_invalid, 0, sizeof(unsigned int), b->stream));
  EventPair evp;
  HIPCHK(hipEventCreate(&evp.e0));
  HIPCHK(hipEventCreate(&evp.e1));
  const int staged = mode & 1, nt = (mode >> 1) & 1;
  for (int i = 0; i < warmup; ++i) if ((rc = rows_launch(b, corrected, layout, staged, nt, r_dev, J_dev)) != EA_OK) return rc;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipLaunchHostFunc(b->stream, [](void *) { std::this_thread::sleep_for(std::chrono::milliseconds(3)); }, nullptr));
  HIPCHK(hipEventRecord(evp.e0, b->stream));
  for (int i = 0; i < launches; ++i) if ((rc = rows_launch(b, corrected, layout, staged, nt, r_dev, J_dev)) != EA_OK) return rc;
  HIPCHK(hipEventRecord(evp.e1, b->stream));
  HIPCHK(hipEventSynchronize(evp.e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, evp.e0, evp.e1));
  *ms_per_launch = (double)ms / launches;
  return EA_OK;
}

extern "C" int ea_batch_set_tuning(ea_batch *b, const char *key, int value) {
  if (!b || !key) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  const std::string k(key);
  if (k == "lds_bytes") b->t_lds_bytes = value;
  else if (k == "points_per_thread") b->t_ppt = value;
  else if (k == "use_lds") b->t_use_lds = value;
  else if (k == "xcd_remap") b->t_xcd = value;
  else if (k == "threads") b->t_nt = value;
  else if (k == "buffer_loads") b->t_buf = value;
  else if (k == "wide_accumulate") b->t_wide = value;
  else if (k == "dt_f32") b->t_img32 = value;
  else if (k == "test_stall_ms") { b->t_test_stall_ms = value; return EA_OK; }
  else if (k == "test_fail_build") { b->t_test_fail_build = value; return EA_OK; }
  else if (k == "solve_streams") { b->t_streams = value; return EA_OK; }
  else if (k == "rows_staged") { b->t_rows_staged = value; return EA_OK; }
  else if (k == "rows_nontemporal") { b->t_rows_nt = value; return EA_OK; }
  else if (k == "poll_results") { b->t_poll = value != 0; return EA_OK; }
  else if (k == "fused_iterations") { b->t_fused = value; return EA_OK; }
  else if (k == "zero_copy_poses") { b->t_zero_copy = value; return EA_OK; }
  else if (k == "poses_per_launch") { b->t_kp_G = value > 0 ? value : 0; b->kp_K = 0; return EA_OK; }  // (resident poses are dropped)
  else return fail(EA_ERR_INVALID_ARG, "unknown tuning key: " + k);
  b->built = false;
  return EA_OK;
}

extern "C" int ea_batch_get_info(const ea_batch *b, const char *key, int64_t *value) {
  if (!b || !key || !value) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  const std::string k(key);
  if (k == "num_tiles") *value = b->ntiles;
  else if (k == "points_per_thread") *value = b->ppt;
  else if (k == "lds_bytes") *value = b->lds_bytes;
  else if (k == "xcd_remap") *value = b->xcd_remap;
  else if (k == "chunk") *value = b->chunk;
  else if (k == "threads") *value = b->nt;
  else if (k == "buffer_loads") *value = b->buffer_loads;
  else if (k == "wide_accumulate") *value = b->wide;
  else if (k == "dt_f32") *value = b->img32;
  else if (k == "num_points") { int64_t s = 0; for (auto *p : b->probs) s += p->n; *value = s; }
  else if (k == "num_rows") *value = b->total_rows;
  else if (k == "poses_per_launch") *value = b->kp_G;  // G of the last ea_batch_set_poses (0: none resident)
  else if (k == "poses_points_per_thread") *value = b->kp_ppt;   // launch shape of the pose-batched evaluation
  else if (k == "poses_threads") *value = b->kp_nt;
  else if (k == "poses_tiles") *value = b->kp_ntiles;            // partial rows (= workgroups with work) per pose
  else if (k == "fused_iterations") *value = b->last_fused;      // the last solve ran one launch per LM iteration (ea_lm_iter_kernel)
  else return fail(EA_ERR_INVALID_ARG, "unknown info key: " + k);
  return EA_OK;
}

// ---- single-problem conveniences = batch of one ------------------------------------------------

extern "C" int ea_batch_row_offsets(ea_batch *b, int64_t *offsets);
extern "C" int ea_batch_eval_rows(ea_batch *b, const double *q, const double *t, int corrected, int layout, void *r_host,
                                  void *J_host, int64_t capacity_rows, int64_t *n_invalid);
extern "C" int ea_batch_eval_rows_device(ea_batch *b, const double *q, const double *t, int corrected, int layout, void *r_dev,
                                         void *J_dev, int64_t capacity_rows, int64_t *n_invalid);

static int self_batch(ea_problem *p, ea_batch **out) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL problem");
  if (!p->self) {
    ea_batch *b = nullptr;
    int rc = ea_batch_create(&b, &p, 1);
    if (rc != EA_OK) return rc;
    p->self = b;
  }
  *out = p->self;
  return EA_OK;
}

extern "C" int ea_eval(ea_problem *p, const double q[4], const double t[3], double *cost, double JtJ[36],
                       double Jtr[6], int64_t *n_invalid) {
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  return ea_batch_eval(b, q, t, cost, JtJ, Jtr, n_invalid);
}

// materialised mode of one problem (its terms included, in term order): rows = ea_problem_num_rows(p)
extern "C" int ea_problem_num_rows(ea_problem *p, int64_t *rows) {
  if (!rows) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  int64_t off[2];
  rc = ea_batch_row_offsets(b, off);
  if (rc == EA_OK) *rows = off[1];
  return rc;
}

extern "C" int ea_eval_rows(ea_problem *p, const double q[4], const double t[3], int corrected, int layout, void *r_host,
                            void *J_host, int64_t capacity_rows, int64_t *n_invalid) {
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  return ea_batch_eval_rows(b, q, t, corrected, layout, r_host, J_host, capacity_rows, n_invalid);
}

extern "C" int ea_eval_rows_device(ea_problem *p, const double q[4], const double t[3], int corrected, int layout, void *r_dev,
                                   void *J_dev, int64_t capacity_rows, int64_t *n_invalid) {
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  return ea_batch_eval_rows_device(b, q, t, corrected, layout, r_dev, J_dev, capacity_rows, n_invalid);
}

extern "C" int ea_cost(ea_problem *p, const double q[4], const double t[3], double *cost, int64_t *n_invalid) {
  return ea_eval(p, q, t, cost, nullptr, nullptr, n_invalid);
}

// the reference's integer-pixel cost report (standalone_edge_align.cpp:2494-2567, :2704-2776)
extern "C" int ea_problem_pixel_cost(ea_problem *p, const double q[4], const double t[3], ea_pixel_cost *out) {
  if (!q || !t || !out) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  rc = batch_build(b);
  if (rc != EA_OK) return rc;
  std::memset(out, 0, sizeof(*out));
  out->max_cost = -1.0;
  if (p->n == 0) return EA_OK;
  rc = batch_upload_poses(b, q, t);
  if (rc != EA_OK) return rc;
  struct Partial { double sum, max, max_u, max_v; long long max_index, inside, outside, pad_; };
  const int nwg = (int)((p->n + kBlockThreads - 1) / kBlockThreads);
  DevBuf buf;
  HIPCHK(cached_malloc(&buf.p, (size_t)nwg * sizeof(Partial), p->device));
  HIPCHK(launch_pixel_cost(p->dtype, b->d_probs, 0, (int)p->n, b->d_poses, buf.p, b->stream));
  std::vector<Partial> parts((size_t)nwg);
  HIPCHK(hipMemcpyAsync(parts.data(), buf.p, (size_t)nwg * sizeof(Partial), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  long long best = 0x7fffffffffffffffLL;
  for (const Partial &w : parts) {  // fixed order: workgroup 0, 1, ...
    out->total_cost += w.sum;
    out->count += w.inside;
    out->outside += w.outside;
    if (w.max > out->max_cost || (w.max == out->max_cost && w.max_index < best)) {
      out->max_cost = w.max; out->max_pixel[0] = w.max_u; out->max_pixel[1] = w.max_v; best = w.max_index;
    }
  }
  out->mean_cost = out->count > 0 ? out->total_cost / (double)out->count : 0.0;
  return EA_OK;
}

extern "C" int ea_eval_points(ea_problem *p, const double q[4], const double t[3], double *r, double *J, int corrected) {
  if (!q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  rc = batch_build(b);
  if (rc != EA_OK) return rc;
  if (p->n == 0) return EA_OK;
  rc = batch_upload_poses(b, q, t);
  if (rc != EA_OK) return rc;
  DevBuf buf_r, buf_J;
  if (r) HIPCHK(cached_malloc(&buf_r.p, p->n * sizeof(double), p->device));
  if (J) HIPCHK(cached_malloc(&buf_J.p, p->n * 6 * sizeof(double), p->device));
  double *d_r = buf_r.as<double>(), *d_J = buf_J.as<double>();
  hipError_t e = launch_eval_points(p->dtype, b->d_probs, 0, (int)p->n, b->d_poses, d_r, d_J, corrected, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (p->order.empty()) {
    if (e == hipSuccess && r) e = hipMemcpy(r, d_r, p->n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && J) e = hipMemcpy(J, d_J, p->n * 6 * sizeof(double), hipMemcpyDeviceToHost);
  } else {  // stored in tile order: hand the rows back in the caller's order
    std::vector<double> tmp((size_t)p->n * 6);
    if (e == hipSuccess && r) {
      e = hipMemcpy(tmp.data(), d_r, p->n * sizeof(double), hipMemcpyDeviceToHost);
      if (e == hipSuccess)
        for (int64_t i = 0; i < p->n; ++i) r[p->order[(size_t)i]] = tmp[(size_t)i];
    }
    if (e == hipSuccess && J) {
      e = hipMemcpy(tmp.data(), d_J, p->n * 6 * sizeof(double), hipMemcpyDeviceToHost);
      if (e == hipSuccess)
        for (int64_t i = 0; i < p->n; ++i)
          for (int a = 0; a < 6; ++a) J[(size_t)p->order[(size_t)i] * 6 + a] = tmp[(size_t)i * 6 + a];
    }
  }
  if (e != hipSuccess) return fail(EA_ERR_HIP, hipGetErrorString(e));
  return EA_OK;
}

extern "C" int ea_solve(ea_problem *p, const ea_options *opt, double q[4], double t[3], ea_summary *summary) {
  ea_batch *b;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  return ea_batch_solve(b, opt, q, t, summary);
}

// ---- one problem sharded by points over several processes / GPUs ----------------------------------
// Every rank holds a shard of the edge points and the whole DT image.  Per trust-region iteration: the local fused
// evaluation (device) -> the 32 accumulator slots -> `allreduce` (in-place sum over ranks: RCCL through
// torch.distributed in edge_alignment_amd/dist.py) -> the same state machine on every rank.  The step is a
// deterministic function of the reduced sums, so the ranks stay in lockstep without a broadcast.  The state machine
// runs on the host here (ea_lm.h, the code the device kernels run): the collective returns its result to the host
// anyway.  A rank with an empty shard takes part with zero sums.
extern "C" int ea_solve_sharded(ea_problem *p, const ea_options *opt_in, ea_allreduce_fn allreduce, void *user, double q[4],
                                double t[3], ea_summary *summary) {
  if (!p || !allreduce || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  const auto t0 = std::chrono::steady_clock::now();
  ea_options o;
  if (opt_in) o = *opt_in; else ea_default_options(&o);
  if (int vrc = check_options(o)) return vrc;
  ea_batch *b = nullptr;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  rc = batch_build(b);
  if (rc != EA_OK) return rc;
  LMOptions lo;
  lo.max_num_iterations = o.max_num_iterations;
  lo.function_tolerance = o.function_tolerance;
  lo.gradient_tolerance = o.gradient_tolerance;
  lo.parameter_tolerance = o.parameter_tolerance;
  lo.initial_trust_region_radius = o.initial_trust_region_radius;
  lo.max_trust_region_radius = o.max_trust_region_radius;
  lo.min_trust_region_radius = o.min_trust_region_radius;
  lo.min_relative_decrease = o.min_relative_decrease;
  lo.min_lm_diagonal = o.min_lm_diagonal;
  lo.max_lm_diagonal = o.max_lm_diagonal;
  lo.max_num_consecutive_invalid_steps = o.max_num_consecutive_invalid_steps;
  lo.jacobi_scaling = o.jacobi_scaling;
  lo.strategy = o.strategy;
  LMState st;
  LMCold cold;
  std::memset(&cold, 0, sizeof(cold));
  std::vector<LMTrace> trv(1);
  LMTrace &tr = trv[0];
  std::memset(&tr, 0, sizeof(tr));
  lm_init(&st, &lo, q, t, p->rot_transposed);
  int guard = o.max_num_iterations + 4;
  while (st.running && guard-- > 0) {
    const double *pose = st.num_evals == 0 ? st.x : st.cand;
    double acc[kAccSlots];
    if (p->n > 0 || !p->terms.empty()) {
      rc = batch_upload_poses(b, pose, pose + 4);
      if (rc != EA_OK) return rc;
      rc = batch_launch_eval(b);
      if (rc != EA_OK) return rc;
      HIPCHK(launch_reduce(b->d_groups, 1, b->d_partials, b->dv_out, b->stream));  // (straight into pinned host memory)
      HIPCHK(hipStreamSynchronize(b->stream));
      std::memcpy(acc, b->h_out[0].acc, sizeof(acc));
    } else {
      std::memset(acc, 0, sizeof(acc));
    }
    if (allreduce(acc, kAccSlots, user) != 0) return fail(EA_ERR_STATE, "the all-reduce callback reported a failure");
    LMPending pend;
    if (st.num_evals == 0) lm_begin_rt(&st, &cold, &tr, &lo, acc, &pend);
    else lm_advance_rt(&st, &cold, &tr, &lo, acc, &pend);
    lm_flush(&pend, &cold, &tr, acc);
  }
  if (st.running) return fail(EA_ERR_STATE, "sharded solve did not terminate");
  for (int k = 0; k < 4; ++k) q[k] = st.x[k];
  for (int k = 0; k < 3; ++k) t[k] = st.x[4 + k];
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int64_t npts = p->n;
  for (ea_problem *tm : p->terms) npts += tm->n;
  if (summary) fill_summary(st, tr, npts, ms, summary);
  return EA_OK;
}

// The point-sharded solve with the exchange kept on the stream: evaluation -> fold into the
// caller's device buffer -> the caller's collective, enqueued on the batch's stream -> the device step kernel reading
// that buffer as "one partial row".  The state machine is the device one of ea_solve; the host only enqueues rounds of
// iterations and looks at the pinned progress word between rounds.  Every rank sees the same sums, hence the same
// state after every iteration, hence stops after the same round: the collectives match up without any extra
// communication.
extern "C" int ea_solve_sharded_device(ea_problem *p, const ea_options *opt_in, ea_device_allreduce_fn allreduce, void *user,
                                       double *device_sums, double q[4], double t[3], ea_summary *summary) {
  if (!p || !allreduce || !device_sums || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (reinterpret_cast<uintptr_t>(device_sums) % 16 != 0) return fail(EA_ERR_INVALID_ARG, "device_sums must be 16-byte aligned");
  const auto t0 = std::chrono::steady_clock::now();
  ea_options o;
  if (opt_in) o = *opt_in; else ea_default_options(&o);
  if (int vrc = check_options(o)) return vrc;
  ea_batch *b = nullptr;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  LMOptions lo;
  lo.max_num_iterations = o.max_num_iterations;
  lo.function_tolerance = o.function_tolerance;
  lo.gradient_tolerance = o.gradient_tolerance;
  lo.parameter_tolerance = o.parameter_tolerance;
  lo.initial_trust_region_radius = o.initial_trust_region_radius;
  lo.max_trust_region_radius = o.max_trust_region_radius;
  lo.min_trust_region_radius = o.min_trust_region_radius;
  lo.min_relative_decrease = o.min_relative_decrease;
  lo.min_lm_diagonal = o.min_lm_diagonal;
  lo.max_lm_diagonal = o.max_lm_diagonal;
  lo.max_num_consecutive_invalid_steps = o.max_num_consecutive_invalid_steps;
  lo.jacobi_scaling = o.jacobi_scaling;
  lo.strategy = o.strategy;
  SolveRun r;
  r.b = b;
  rc = solve_start(r, o, lo, q, t);  // builds the batch, uploads pose + state, arms the progress words
  if (rc != EA_OK) return rc;
  if (!b->d_one_row) {
    const GroupDesc one = {0, 1, 0, 1};
    HIPCHK(hipMalloc(&b->d_one_row, sizeof(GroupDesc)));
    HIPCHK(hipMemcpy(b->d_one_row, &one, sizeof(one), hipMemcpyHostToDevice));
  }
  // Look-ahead rule.  The host keeps `ahead` iterations queued; iteration i + ahead is enqueued once iteration i is
  // COMPLETE (the step kernel posts that behind its flag) unless the solve had finished by iteration i.  The decision
  // depends on (i, the iteration the solve finished at) only -- quantities every rank agrees on, whenever its host happens
  // to look -- so every rank enqueues exactly (finishing iteration + ahead) iterations and the collectives match up
  // without the ranks talking about it.  The device never waits for the host: the next iteration is in the queue while
  // this one runs.  (Round 2 enqueued rounds of four behind an event wait: the device idled while the host looked and
  // enqueued, 5.1e4 against 8.1e4 iterations/s unsharded on one rank.)
  const int ahead = o.iterations_per_sync > 0 ? o.iterations_per_sync : 2;
  const double timeout_ms = resolve_timeout_ms(o);
  auto enqueue_iteration = [&]() -> int {
    int rc2 = batch_launch_eval(b);  // (an empty shard launches nothing; its fold below yields zeros)
    if (rc2 != EA_OK) return rc2;
    HIPCHK(launch_reduce(b->d_groups, 1, b->d_partials, reinterpret_cast<EvalOut *>(device_sums), b->stream));
    if (allreduce(device_sums, kAccSlots, (void *)b->stream, user) != 0) {
      b->needs_drain = true;
      return fail(EA_ERR_STATE, "the all-reduce callback reported a failure");
    }
    HIPCHK(launch_lm_step(b->d_one_row, 1, device_sums, b->d_poses, b->d_states, b->d_cold, b->d_traces, lo, b->d_progress,
                          b->dv_states, b->dv_traces, GroupDesc{0, 1, 0, 1}, /*post_done=*/1, b->stream));
    ++r.enq;
    return EA_OK;
  };
  for (int k = 0; k < ahead && r.enq < r.budget; ++k)
    if ((rc = enqueue_iteration()) != EA_OK) return rc;
  bool finished = false;
  for (int i = 0; i < r.enq; ++i) {
    SpinWait wait(timeout_ms);
    int seen = -1;
    while (__atomic_load_n(&b->h_progress[2], __ATOMIC_ACQUIRE) < i + 1) {
      const int started = __atomic_load_n(&b->h_progress[1], __ATOMIC_ACQUIRE);
      if (started != seen) { seen = started; wait.progress(); }
      else if (wait.poll()) {
        b->needs_drain = true;
        return fail(EA_ERR_HIP, "sharded solve deadline: no progress on the device (is every rank taking part in the collective?)");
      }
    }
    // the flag as iteration i left it; the final state was delivered in front of it
    if (__atomic_load_n(&b->h_progress[0], __ATOMIC_ACQUIRE) == 0 && b->hd_states[0].num_evals <= i + 1) { finished = true; break; }
    if (r.enq < r.budget && (rc = enqueue_iteration()) != EA_OK) return rc;
  }
  // the iterations queued past the end find the solve finished and return at once, their collectives still run (on every
  // rank alike); the caller's buffer must outlive them
  HIPCHK(hipStreamSynchronize(b->stream));
  r.fetch = !finished;
  r.done = true;
  rc = solve_collect(r, o, summary != nullptr);
  if (rc != EA_OK) return rc;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  solve_report(r, o, ms, q, t, summary);
  return EA_OK;
}

// Coarse-to-fine driver (BASELINE config C3; the reference has no pyramid -- SURVEY 8f row 4): levels[0] is the finest
// level; the solve starts on levels[nlevels-1] and carries the pose down level by level.  Every level is a complete
// problem (its own points, DT image and intrinsics scaled by the caller).  A level that fails (termination FAILURE)
// stops the descent and its status is returned through the summaries; q, t hold the last pose reached.
// The point-sharded solve in the one-launch-per-iteration form (ea_solve_sharded_comm).  Between evaluation and step sits the
// exchange; with ea_lm_iter_kernel every workgroup folds the rows itself, so what is exchanged is the ROWS: launch j
// evaluates this rank's shard into its rows, ONE in-place all-reduce sums every rank's rows (a few tens of KB instead of 256
// bytes -- the same latency-bound collective), launch j + 1 folds the summed rows, steps and evaluates.  Per iteration: one
// kernel launch + one collective, where the (evaluate, fold, all-reduce, step) form has three launches + one collective.
// Every rank folds the same bits in the same order, so the replicated state machines stay in lockstep as before.
// Ranks may hold different numbers of rows (shards differ by a point): `agree` takes {this rank cannot, its row count} to the
// maximum over the ranks; every rank folds the widest count, its own rows beyond its shard kept at zero.
// *used = 0: some rank's shard does not qualify (see solve_start) -- the caller falls back to ea_solve_sharded_device.
extern "C" int ea_internal_solve_sharded_rows(ea_problem *p, const ea_options *opt_in, ea_device_allreduce_fn allreduce,
                                              int (*agree)(int vals[2], void *user), void *user, double q[4], double t[3],
                                              ea_summary *summary, int *used) {
  if (!p || !allreduce || !agree || !q || !t || !used) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  *used = 0;
  const auto t0 = std::chrono::steady_clock::now();
  ea_options o;
  if (opt_in) o = *opt_in; else ea_default_options(&o);
  if (int vrc = check_options(o)) return vrc;
  ea_batch *b = nullptr;
  int rc = self_batch(p, &b);
  if (rc != EA_OK) return rc;
  LMOptions lo;
  lo.max_num_iterations = o.max_num_iterations;
  lo.function_tolerance = o.function_tolerance;
  lo.gradient_tolerance = o.gradient_tolerance;
  lo.parameter_tolerance = o.parameter_tolerance;
  lo.initial_trust_region_radius = o.initial_trust_region_radius;
  lo.max_trust_region_radius = o.max_trust_region_radius;
  lo.min_trust_region_radius = o.min_trust_region_radius;
  lo.min_relative_decrease = o.min_relative_decrease;
  lo.min_lm_diagonal = o.min_lm_diagonal;
  lo.max_lm_diagonal = o.max_lm_diagonal;
  lo.max_num_consecutive_invalid_steps = o.max_num_consecutive_invalid_steps;
  lo.jacobi_scaling = o.jacobi_scaling;
  lo.strategy = o.strategy;
  SolveRun r;
  r.b = b;
  rc = solve_start(r, o, lo, q, t);  // builds the batch, uploads pose + state, arms the progress words, decides r.fused
  if (rc != EA_OK) return rc;
  int vals[2] = {r.fused ? 0 : 1, b->ntiles};
  if (agree(vals, user) != 0) return fail(EA_ERR_STATE, "the ranks could not agree on the shape of the sharded solve");
  if (vals[0] != 0) { HIPCHK(hipStreamSynchronize(b->stream)); return EA_OK; }  // (*used = 0)
  const int max_rows = vals[1];
  if (max_rows > b->tiles_cap || max_rows > b->tiles_cap_alt) {
    HIPCHK(hipStreamSynchronize(b->stream));
    cached_free(b->d_partials); cached_free(b->d_partials_alt);
    b->d_partials = b->d_partials_alt = nullptr;
    b->tiles_cap = b->tiles_cap_alt = 0;
    const int cap = max_rows + 16;
    HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_partials), (size_t)cap * kAccSlots * sizeof(double), b->device));
    b->tiles_cap = cap;
    HIPCHK(cached_malloc(reinterpret_cast<void **>(&b->d_partials_alt), (size_t)cap * kAccSlots * sizeof(double), b->device));
    b->tiles_cap_alt = cap;
  }
  double *rows[2] = {b->d_partials, b->d_partials_alt};
  if (max_rows > b->ntiles)
    for (int k = 0; k < 2; ++k)
      HIPCHK(hipMemsetAsync(rows[k] + (size_t)b->ntiles * kAccSlots, 0, (size_t)(max_rows - b->ntiles) * kAccSlots * sizeof(double), b->stream));
  LMState *st[2] = {b->d_states, reinterpret_cast<LMState *>(b->d_iter_alt)};
  LMCold *cold[2] = {b->d_cold, reinterpret_cast<LMCold *>(b->d_iter_alt + (size_t)b->iter_alt_count * sizeof(LMState))};
  const GroupDesc fold_range = {0, max_rows, 0, 1};
  auto exchange = [&](double *buf) -> int {
    if (allreduce(buf, max_rows * kAccSlots, (void *)b->stream, user) != 0) {
      b->needs_drain = true;
      return fail(EA_ERR_STATE, "the all-reduce callback reported a failure");
    }
    return EA_OK;
  };
  // the look-ahead rule of ea_solve_sharded_device: iteration i + ahead goes out once iteration i is complete unless the solve
  // had finished by iteration i -- every rank enqueues the same number of launches and collectives
  const int ahead = o.iterations_per_sync > 0 ? o.iterations_per_sync : 2;
  const double timeout_ms = resolve_timeout_ms(o);
  rc = batch_launch_eval(b);  // launch 0: the evaluation at the start pose
  if (rc != EA_OK) return rc;
  if ((rc = exchange(rows[0])) != EA_OK) return rc;
  auto enqueue_iteration = [&]() -> int {
    const int in = r.enq & 1, out = in ^ 1;
    HIPCHK(launch_lm_iter(b->dtype, b->ppt, b->d_probs, 1, b->chunk, b->max_chunks, b->xcd_remap, b->d_poses, rows[in], rows[out],
                          b->buffer_loads, b->img32, b->x0, b->y0, b->z0, b->n0, b->d_groups, st[in], st[out], cold[in], cold[out],
                          b->d_traces, lo, b->d_progress, b->dv_states, b->dv_traces, fold_range, /*post_done=*/1, b->stream));
    int rc2 = exchange(rows[out]);
    if (rc2 != EA_OK) return rc2;
    ++r.enq;
    return EA_OK;
  };
  for (int k = 0; k < ahead && r.enq < r.budget; ++k)
    if ((rc = enqueue_iteration()) != EA_OK) return rc;
  bool finished = false;
  for (int i = 0; i < r.enq; ++i) {
    SpinWait wait(timeout_ms);
    int seen = -1;
    while (__atomic_load_n(&b->h_progress[2], __ATOMIC_ACQUIRE) < i + 1) {
      const int started = __atomic_load_n(&b->h_progress[1], __ATOMIC_ACQUIRE);
      if (started != seen) { seen = started; wait.progress(); }
      else if (wait.poll()) {
        b->needs_drain = true;
        return fail(EA_ERR_HIP, "sharded solve deadline: no progress on the device (is every rank taking part in the collective?)");
      }
    }
    if (__atomic_load_n(&b->h_progress[0], __ATOMIC_ACQUIRE) == 0 && b->hd_states[0].num_evals <= i + 1) { finished = true; break; }
    if (r.enq < r.budget && (rc = enqueue_iteration()) != EA_OK) return rc;
  }
  // A finished solve has delivered its result into pinned host memory in front of the flag seen above: return on it.  The
  // launches and collectives queued past the end (every rank alike) drain behind our back -- they touch only buffers the
  // library owns, and whatever uses this stream or the communicator next is ordered behind them (ea_comm_destroy waits for
  // the stream of its last solve).  Only a solve cut short by the launch budget has to wait and fetch.
  if (!finished) HIPCHK(hipStreamSynchronize(b->stream));
  r.fetch = !finished;
  r.done = true;
  rc = solve_collect(r, o, summary != nullptr);
  if (rc != EA_OK) return rc;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  solve_report(r, o, ms, q, t, summary);
  *used = 1;
  return EA_OK;
}

extern "C" int ea_solve_pyramid(ea_problem *const *levels, int nlevels, const ea_options *opt, double q[4], double t[3],
                                ea_summary *summaries) {
  if (!levels || nlevels < 1 || !q || !t) return fail(EA_ERR_INVALID_ARG, "bad argument");
  for (int l = 0; l < nlevels; ++l)
    if (!levels[l]) return fail(EA_ERR_INVALID_ARG, "NULL level");
  for (int l = nlevels - 1; l >= 0; --l) {
    ea_summary local;
    ea_summary *s = summaries ? &summaries[l] : &local;
    const int rc = ea_solve(levels[l], opt, q, t, s);
    if (rc != EA_OK) return rc;
    if (s->termination == EA_FAILURE) {
      for (int k = l - 1; k >= 0 && summaries; --k) std::memset(&summaries[k], 0, sizeof(ea_summary));
      break;
    }
  }
  return EA_OK;
}

static int ref_points_from_last_now(ea_problem *p, int kind, const uint16_t *depth, int height, int width, double z_scaling,
                                    int threshold);  // (with the frame producers below)
struct RefPointsJob;
static int ref_points_begin(ea_problem *p, int kind, const uint16_t *depth, int height, int width, int threshold, RefPointsJob *job);
static int ref_points_finish(ea_problem *p, const RefPointsJob &job, int height, int width, double z_scaling, int threshold);
// what ref_points_begin left in flight on the null stream: the depth frame on its way up and the per-block edge counts
struct RefPointsJob {
  bool started = false;
  uint8_t *d_edges = nullptr;
  uint16_t *d_depth = nullptr;
  int *d_counts = nullptr, *d_total = nullptr;
};

// ---- frame-to-frame driver (the reference aligns one stored pair, src/ea.cpp:155-200) -------------
// Every pushed frame is aligned against the previous one: its DT image is produced, the previous frame's edge points
// are solved against it starting from the last relative pose (constant-velocity prior), then the new frame's edge
// points become the reference.  All of it stays on the device; one ea_problem is reused.
struct ea_tracker {
  ea_problem *p = nullptr;
  int flavour = 0;      // 0: get_aX / get_distance_transform, 1: Canny (get_aX_canny / get_distance_transform2)
  int frames = 0;
  double q[4] = {1, 0, 0, 0}, t[3] = {0, 0, 0};
};

extern "C" int ea_tracker_create(ea_tracker **out, const ea_camera *cam, int dtype, int device, int flavour) {
  if (!out) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (flavour != 0 && flavour != 1) return fail(EA_ERR_INVALID_ARG, "unknown pre-processing flavour");
  ea_tracker *tr = new (std::nothrow) ea_tracker;
  if (!tr) return fail(EA_ERR_ALLOC, "out of host memory");
  const int rc = ea_problem_create(&tr->p, cam, dtype, device);
  if (rc != EA_OK) { delete tr; return rc; }
  tr->flavour = flavour;
  *out = tr;
  return EA_OK;
}

extern "C" void ea_tracker_destroy(ea_tracker *tr) {
  if (!tr) return;
  ea_problem_destroy(tr->p);
  delete tr;
}

extern "C" ea_problem *ea_tracker_problem(ea_tracker *tr) { return tr ? tr->p : nullptr; }

// q_rel, t_rel: pose of the previous frame in the new frame's coordinates (b_T_a with a = previous, b = new); identity
// for the first frame.  aligned (nullable): 1 when a solve took place.  A failed solve keeps the prior for the next frame.
extern "C" int ea_tracker_push_frame(ea_tracker *tr, const uint8_t *bgr, const uint16_t *depth, int height, int width,
                                     double z_scaling, const ea_options *opt, double q_rel[4], double t_rel[3],
                                     ea_summary *summary, int *aligned) {
  if (!tr || !bgr || !depth || !q_rel || !t_rel) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  // every argument is checked before the tracker's problem is touched: a rejected call leaves the tracker as it was
  if (!(z_scaling > 0.0)) return fail(EA_ERR_INVALID_ARG, "z_scaling must be > 0");
  if (height < 1 || width < 1) return fail(EA_ERR_INVALID_ARG, "bad frame extent");
  int rc = EA_OK;
  if (aligned) *aligned = 0;
  if (summary) std::memset(summary, 0, sizeof(*summary));
  double q_new[4], t_new[3];
  std::memcpy(q_new, tr->q, sizeof(q_new));
  std::memcpy(t_new, tr->t, sizeof(t_new));
  RefPointsJob job;
  if (tr->frames > 0 && ea_problem_num_points(tr->p) > 0) {
    rc = tr->flavour == 0 ? ea_problem_set_now_frame(tr->p, bgr, height, width, 35, 1, 1)
                          : ea_problem_set_now_frame_canny(tr->p, bgr, nullptr, height, width, 30, 90, 1, 0.0, 1.0);
    if (rc != EA_OK) return rc;
    // the new frame's depth goes up and its edge pixels are counted WHILE the previous reference is solved against the image
    // just produced (null stream beside the solve's non-blocking stream; neither touches what the other uses)
    (void)ref_points_begin(tr->p, tr->flavour == 0 ? 1 : 2, depth, height, width, tr->flavour == 0 ? 35 : 0, &job);
    double q[4], t[3];
    std::memcpy(q, tr->q, sizeof(q));
    std::memcpy(t, tr->t, sizeof(t));
    ea_summary s;
    rc = ea_solve(tr->p, opt, q, t, &s);
    if (rc != EA_OK) {
      if (job.started) (void)hipDeviceSynchronize();  // (nothing of this frame may stay in flight behind a failed push)
      return rc;
    }
    if (s.termination != EA_FAILURE) {
      std::memcpy(q_new, q, sizeof(q));
      std::memcpy(t_new, t, sizeof(t));
    }
    if (summary) *summary = s;
    if (aligned) *aligned = 1;
  }
  // the frame's edge strength / edge map is still in the workspace when it has just been the "now" frame
  rc = job.started ? ref_points_finish(tr->p, job, height, width, z_scaling, tr->flavour == 0 ? 35 : 0)
                   : ref_points_from_last_now(tr->p, tr->flavour == 0 ? 1 : 2, depth, height, width, z_scaling, tr->flavour == 0 ? 35 : 0);
  if (rc == EA_ERR_STATE)
    rc = tr->flavour == 0 ? ea_problem_set_ref_frame(tr->p, bgr, depth, height, width, z_scaling, 35)
                          : ea_problem_set_ref_frame_canny(tr->p, bgr, depth, height, width, z_scaling, 30, 90);
  if (rc != EA_OK) {
    // the DT image is the new frame's but no reference came out of it: drop the stale reference so that the next push
    // starts a fresh chain instead of aligning frame k-1's points against frame k+2 from an advanced prior
    (void)reserve_points(tr->p, 0);
    tr->p->version++;
    tr->frames = 0;
    return rc;
  }
  // prior and frame count advance only with the new reference in place
  std::memcpy(tr->q, q_new, sizeof(q_new));
  std::memcpy(tr->t, t_new, sizeof(t_new));
  std::memcpy(q_rel, tr->q, sizeof(tr->q));
  std::memcpy(t_rel, tr->t, sizeof(tr->t));
  tr->frames += 1;
  return EA_OK;
}

// ---- self-test of the wavefront reduction primitives (DPP row_mirror / row_half_mirror with bank
// masks, v_permlane16/32_swap, quad_perm): in = 32 slots x 64 lanes (fp32); out32/out64 = the 32 wave
// totals from the fp32 and fp64 reductions; stages (nullable) = 16+8+4+2 rows of 64 lanes.
extern "C" int ea_selftest_wave_reduce(int device, const float *in, double *out32, double *out64, float *stages) {
  if (!in || !out32 || !out64) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  int rc = check_device(device);
  if (rc != EA_OK) return rc;
  HIPCHK(hipSetDevice(device));
  DevBuf b_in, b_st, b_o;
  HIPCHK(hipMalloc(&b_in.p, 32 * 64 * sizeof(float)));
  HIPCHK(hipMalloc(&b_st.p, 30 * 64 * sizeof(float)));
  HIPCHK(hipMalloc(&b_o.p, 64 * sizeof(double)));
  float *d_in = b_in.as<float>(), *d_st = b_st.as<float>();
  double *d_o = b_o.as<double>();
  HIPCHK(hipMemcpy(d_in, in, 32 * 64 * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_o, 0, 64 * sizeof(double)));
  hipError_t e = launch_selftest_reduce(d_in, d_st, d_st + 16 * 64, d_st + 24 * 64, d_st + 28 * 64, d_o, d_o + 32, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out32, d_o, 32 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out64, d_o + 32, 32 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && stages) e = hipMemcpy(stages, d_st, 30 * 64 * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(EA_ERR_HIP, hipGetErrorString(e));
  return EA_OK;
}

// ---- frame producers: raw images -> edge points / DT image, on the device -------

namespace {
struct WsCarver {
  unsigned char *base;
  size_t off = 0;
  template <typename U> U *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    U *r = reinterpret_cast<U *>(base + off);
    off += n * sizeof(U);
    return r;
  }
};
}  // namespace

static int ensure_ws(ea_problem *p, size_t bytes) {
  p->ws_now_kind = 0;  // every producer starts by calling this: whatever the workspace held is about to be overwritten
  if (p->ws_bytes >= bytes) return EA_OK;
  if (p->ws) { cached_free(p->ws); p->ws = nullptr; p->ws_bytes = 0; }
  HIPCHK(cached_malloc(reinterpret_cast<void **>(&p->ws), bytes, p->device));
  p->ws_bytes = bytes;
  return EA_OK;
}

static size_t frame_ws_bytes(int H, int W) {
  const size_t np = (size_t)H * W;
  return np * 3 + np * 2 + np * 3 /*gray, lap, mask*/ + np * 4 * 2 /*G, dist*/ + np * 4 /*plain float*/ +
         np * 4 + np * 4 + np /*Canny: magnitudes, direction / label / edge / keep bytes, mask*/ + 16 * 256 +
         16 * (size_t)((H + 31) / 32 + 1) * W /*segment ends + carries*/ +
         ((np + 1023) / 1024 + 8) * 4 + 64 * 256;
}

static int ref_frame_impl(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, const uint16_t *depth, int height,
                          int width, double z_scaling, int threshold) {
  if (!p || !bgr || !depth) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (height < 3 || width < 3 || (int64_t)height * width > 0x3fffffff) return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  if (!(z_scaling > 0.0)) return fail(EA_ERR_INVALID_ARG, "z_scaling must be > 0");
  HIPCHK(hipSetDevice(p->device));
  int rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  const size_t np = (size_t)height * width;
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  uint16_t *d_depth = ws.take<uint16_t>(np);
  uint8_t *d_gray = ws.take<uint8_t>(np), *d_lap = ws.take<uint8_t>(np);
  uint8_t *d_keep = mask ? ws.take<uint8_t>(np) : nullptr;
  const int nblocks = (int)((np + 1023) / 1024);
  int *d_counts = ws.take<int>(nblocks + 1);
  int *d_total = d_counts + nblocks;
  HIPCHK(hipMemcpyAsync(d_bgr, bgr, np * 3, hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipMemcpyAsync(d_depth, depth, np * 2, hipMemcpyHostToDevice, nullptr));
  HIPCHK(launch_edge_strength(d_bgr, height, width, d_gray, d_lap, nullptr));
  if (mask) {
    HIPCHK(hipMemcpyAsync(d_keep, mask, np, hipMemcpyHostToDevice, nullptr));
    HIPCHK(launch_gate_by_mask(d_lap, d_keep, height, width, nullptr));
  }
  HIPCHK(launch_edge_count_scan(d_lap, d_depth, height, width, threshold, d_counts, d_total, nullptr));
  int total = 0;
  HIPCHK(hipMemcpy(&total, d_total, sizeof(int), hipMemcpyDeviceToHost));
  p->version++;
  rc = reserve_points(p, total);
  if (rc != EA_OK) return rc;
  if (total > 0) {
    HIPCHK(launch_edge_scatter(p->dtype, d_lap, d_depth, height, width, threshold, d_counts, p->cam.fx, p->cam.fy, p->cam.cx,
                               p->cam.cy, z_scaling, p->d_x, p->d_y, p->d_z, total, nullptr));
    HIPCHK(hipDeviceSynchronize());
  }
  p->n = total;
  return EA_OK;
}

// Edge points of the frame the last set_now_frame[_canny] call processed, from what that call left in the workspace
// (Laplacian strength / Canny edge map): only the depth image goes up, no second upload or filtering of the colour
// frame.  Same thresholds, same compaction and back-projection as ea_problem_set_ref_frame[_canny] => the same points.
// Returns EA_ERR_STATE when the workspace does not hold that frame (the caller then takes the full path).
// The new reference frame's edge points from what the "now" producer left in the workspace, in two halves so that the tracker
// can put the first -- depth upload and per-block edge counts, both asynchronous on the null stream -- in front of the solve of
// the previous reference (which runs on the batch's own non-blocking stream and touches neither the workspace nor the depth)
// and the second -- count read-back, compaction -- behind it.
static int ref_points_begin(ea_problem *p, int kind, const uint16_t *depth, int height, int width, int threshold, RefPointsJob *job) {
  job->started = false;
  if (p->ws_now_kind != kind || p->ws_now_h != height || p->ws_now_w != width || (int64_t)height * width < 4096)
    return EA_ERR_STATE;
  HIPCHK(hipSetDevice(p->device));
  const size_t np = (size_t)height * width;
  // the carve of the producer that ran, to find its buffers again
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  uint8_t *d_gray = ws.take<uint8_t>(np);
  uint8_t *d_edges;
  if (kind == 1) {
    d_edges = ws.take<uint8_t>(np);  // d_lap
  } else {
    (void)ws.take<int>(np);      // magnitudes
    (void)ws.take<uint8_t>(np);  // direction classes
    (void)ws.take<uint8_t>(np);  // labels
    d_edges = ws.take<uint8_t>(np);
  }
  // the colour frame and its gray version are not needed any more: depth and the block counts take their place
  uint16_t *d_depth = reinterpret_cast<uint16_t *>(d_bgr);
  const int nblocks = (int)((np + 1023) / 1024);
  int *d_counts = reinterpret_cast<int *>(d_gray);
  int *d_total = d_counts + nblocks;
  p->ws_now_kind = 0;
  HIPCHK(hipMemcpyAsync(d_depth, depth, np * 2, hipMemcpyHostToDevice, nullptr));
  HIPCHK(launch_edge_count_scan(d_edges, d_depth, height, width, threshold, d_counts, d_total, nullptr));
  job->started = true;
  job->d_edges = d_edges; job->d_depth = d_depth; job->d_counts = d_counts; job->d_total = d_total;
  return EA_OK;
}

static int ref_points_finish(ea_problem *p, const RefPointsJob &job, int height, int width, double z_scaling, int threshold) {
  if (!job.started) return EA_ERR_STATE;
  int total = 0;
  HIPCHK(hipMemcpy(&total, job.d_total, sizeof(int), hipMemcpyDeviceToHost));
  p->version++;
  int rc = reserve_points(p, total);
  if (rc != EA_OK) return rc;
  if (total > 0) {
    HIPCHK(launch_edge_scatter(p->dtype, job.d_edges, job.d_depth, height, width, threshold, job.d_counts, p->cam.fx, p->cam.fy, p->cam.cx,
                               p->cam.cy, z_scaling, p->d_x, p->d_y, p->d_z, total, nullptr));
    HIPCHK(hipDeviceSynchronize());
  }
  p->n = total;
  return EA_OK;
}

static int ref_points_from_last_now(ea_problem *p, int kind, const uint16_t *depth, int height, int width, double z_scaling,
                                    int threshold) {
  RefPointsJob job;
  int rc = ref_points_begin(p, kind, depth, height, width, threshold, &job);
  if (rc != EA_OK) return rc;
  return ref_points_finish(p, job, height, width, z_scaling, threshold);
}

extern "C" int ea_problem_set_ref_frame(ea_problem *p, const uint8_t *bgr, const uint16_t *depth, int height, int width,
                                        double z_scaling, int threshold) {
  return ref_frame_impl(p, bgr, nullptr, depth, height, width, z_scaling, threshold);
}

// get_aX_mask (ref: utils.cpp:283-369, call sites standalone_edge_align.cpp:1039, :1081): also requires mask > 0
extern "C" int ea_problem_set_ref_frame_masked(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, const uint16_t *depth,
                                               int height, int width, double z_scaling, int threshold) {
  if (!mask) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  return ref_frame_impl(p, bgr, mask, depth, height, width, z_scaling, threshold);
}

// mask (0 = edge / DT source) -> chamfer DT -> [normalise to [lo, hi]] -> the problem's padded DT image
static int dt_from_mask(ea_problem *p, WsCarver &ws, const uint8_t *d_mask, int height, int width, int normalize,
                        double lo, double hi, int **dist_out, float **plain_out, bool precise = false) {
  // the row pass stages one image row of column distances in LDS (4 bytes per pixel, 64 KB)
  if (width > 16384) return fail(EA_ERR_INVALID_ARG, "frames wider than 16384 pixels are not supported by the DT producers");
  const size_t np = (size_t)height * width;
  int *d_G = ws.take<int>(np), *d_dist = ws.take<int>(np);  // d_dist doubles as the float32 distance when `precise`
  int *d_scan = ws.take<int>(4 * (size_t)((height + 31) / 32) * width);
  float *d_plain = ws.take<float>(np);
  unsigned int *d_minmax = ws.take<unsigned int>(2);
  float *d_dist_f32 = precise ? reinterpret_cast<float *>(d_dist) : nullptr;
  HIPCHK(launch_chamfer(d_mask, height, width, d_G, d_scan, d_dist, d_dist_f32, d_minmax, nullptr));
  {
    int rc = alloc_dt(p, width, height);
    if (rc != EA_OK) return rc;
  }
  HIPCHK(launch_dt_store(p->dtype, d_dist, d_dist_f32, height, width, d_minmax, normalize, lo, hi, p->d_dt, p->pitch, d_plain,
                         p->d_dt32, nullptr));
  HIPCHK(hipDeviceSynchronize());
  p->dt32_exact = p->d_dt32 != nullptr;  // the producers compute the distance transform in float32, as OpenCV does
  p->version++;
  if (dist_out) *dist_out = d_dist;
  if (plain_out) *plain_out = d_plain;
  return EA_OK;
}

static int run_dt(ea_problem *p, WsCarver &ws, const uint8_t *d_bgr, int height, int width, int threshold, int median,
                  int normalize, uint8_t **lap_out, uint8_t **mask_out, int **dist_out, float **plain_out) {
  const size_t np = (size_t)height * width;
  uint8_t *d_gray = ws.take<uint8_t>(np), *d_lap = ws.take<uint8_t>(np), *d_mask = ws.take<uint8_t>(np);
  HIPCHK(launch_edge_strength(d_bgr, height, width, d_gray, d_lap, nullptr));
  HIPCHK(launch_threshold_median(d_lap, height, width, threshold, median, d_mask, nullptr));
  if (lap_out) *lap_out = d_lap;
  if (mask_out) *mask_out = d_mask;
  return dt_from_mask(p, ws, d_mask, height, width, normalize, 0.0, 1.0, dist_out, plain_out);
}

// cv::Canny's integer thresholds (L1 magnitude): floor of the ordered pair.  NaN is refused; values beyond any magnitude an
// 8-bit image can produce (|dx| + |dy| <= 2040) are clamped before the conversion, which is undefined for them otherwise.
static int canny_thresholds(double t1, double t2, int *low, int *high) {
  if (t1 != t1 || t2 != t2) return fail(EA_ERR_INVALID_ARG, "Canny threshold is NaN");
  const double lo = std::min(t1, t2), hi = std::max(t1, t2);
  *low = (int)std::floor(std::min(std::max(lo, -1e9), 1e9));
  *high = (int)std::floor(std::min(std::max(hi, -1e9), 1e9));
  return EA_OK;
}

// blur 3x3 -> gray -> Canny(low, high) [-> AND (keep > 1)]: edge map and its inverse in the workspace
static int run_canny(WsCarver &ws, const uint8_t *d_bgr, const uint8_t *d_keep, int height, int width, int low, int high,
                     uint8_t **edges_out, uint8_t **inv_out, int *rounds_out, int l2_bgr = 0) {
  const size_t np = (size_t)height * width;
  uint8_t *d_gray = ws.take<uint8_t>(np);
  int *d_mag = ws.take<int>(np);
  uint8_t *d_dir = ws.take<uint8_t>(np), *d_label = ws.take<uint8_t>(np);
  uint8_t *d_edges = ws.take<uint8_t>(np), *d_inv = ws.take<uint8_t>(np);
  int *d_changed = ws.take<int>(16);  // one change flag per launch of a hysteresis batch
  HIPCHK(launch_canny(d_bgr, height, width, low, high, l2_bgr, d_keep, d_gray, d_mag, d_dir, d_label, d_edges, d_inv,
                      d_changed, rounds_out, nullptr));
  *edges_out = d_edges;
  *inv_out = d_inv;
  return EA_OK;
}

static int check_frame_args(const ea_problem *p, const void *bgr, int height, int width) {
  if (!p || !bgr) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (height < 3 || width < 3 || height > 32768 || width > 32768 || (int64_t)height * width > 0x3fffffff)
    return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  return EA_OK;
}

// Canny flavour of the reference frame: get_aX_canny (ref: utils.cpp:371-462)
extern "C" int ea_problem_set_ref_frame_canny(ea_problem *p, const uint8_t *bgr, const uint16_t *depth, int height,
                                              int width, double z_scaling, double low_threshold, double high_threshold) {
  int rc = check_frame_args(p, bgr, height, width);
  if (rc != EA_OK) return rc;
  if (!depth) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (!(z_scaling > 0.0) || !(z_scaling <= DBL_MAX)) return fail(EA_ERR_INVALID_ARG, "z_scaling must be > 0 and finite");
  int lo, hi;
  rc = canny_thresholds(low_threshold, high_threshold, &lo, &hi);
  if (rc != EA_OK) return rc;
  HIPCHK(hipSetDevice(p->device));
  rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  const size_t np = (size_t)height * width;
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  uint16_t *d_depth = ws.take<uint16_t>(np);
  const int nblocks = (int)((np + 1023) / 1024);
  int *d_counts = ws.take<int>(nblocks + 1);
  int *d_total = d_counts + nblocks;
  HIPCHK(hipMemcpyAsync(d_bgr, bgr, np * 3, hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipMemcpyAsync(d_depth, depth, np * 2, hipMemcpyHostToDevice, nullptr));
  uint8_t *d_edges, *d_inv;
  rc = run_canny(ws, d_bgr, nullptr, height, width, lo, hi, &d_edges, &d_inv, nullptr);
  if (rc != EA_OK) return rc;
  // ref: utils.cpp:441 -- all_grad(i) > 0 && Z > 0 on the 0/255 edge map
  HIPCHK(launch_edge_count_scan(d_edges, d_depth, height, width, 0, d_counts, d_total, nullptr));
  int total = 0;
  HIPCHK(hipMemcpy(&total, d_total, sizeof(int), hipMemcpyDeviceToHost));
  p->version++;
  rc = reserve_points(p, total);
  if (rc != EA_OK) return rc;
  if (total > 0) {
    HIPCHK(launch_edge_scatter(p->dtype, d_edges, d_depth, height, width, 0, d_counts, p->cam.fx, p->cam.fy, p->cam.cx,
                               p->cam.cy, z_scaling, p->d_x, p->d_y, p->d_z, total, nullptr));
    HIPCHK(hipDeviceSynchronize());
  }
  p->n = total;
  return EA_OK;
}

// Canny flavour of the current frame: get_distance_transform2 / _masked / _NoNormalize / _masked_NoNormalize
// (ref: utils.cpp:85-199).  mask (nullable): H x W bytes, edges survive where mask > 1.  normalize != 0: min-max to
// [norm_lo, norm_hi] ((0,1) at :103, (0,255) at :138).  The debug outputs may be NULL.
static int now_frame_canny(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, int height, int width, double low_threshold,
                           double high_threshold, int normalize, double norm_lo, double norm_hi, uint8_t *edges_out,
                           int32_t *chamfer_fix_out, float *dt_out, int *rounds_out) {
  int rc = check_frame_args(p, bgr, height, width);
  if (rc != EA_OK) return rc;
  int lo, hi;
  rc = canny_thresholds(low_threshold, high_threshold, &lo, &hi);
  if (rc != EA_OK) return rc;
  if (normalize && (!(std::fabs(norm_lo) <= DBL_MAX) || !(std::fabs(norm_hi) <= DBL_MAX)))
    return fail(EA_ERR_INVALID_ARG, "normalisation range must be finite");
  HIPCHK(hipSetDevice(p->device));
  rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  const size_t np = (size_t)height * width;
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  uint8_t *d_keep = mask ? ws.take<uint8_t>(np) : nullptr;
  HIPCHK(hipMemcpyAsync(d_bgr, bgr, np * 3, hipMemcpyHostToDevice, nullptr));
  if (mask) HIPCHK(hipMemcpyAsync(d_keep, mask, np, hipMemcpyHostToDevice, nullptr));
  uint8_t *d_edges, *d_inv;
  rc = run_canny(ws, d_bgr, d_keep, height, width, lo, hi, &d_edges, &d_inv, rounds_out);
  if (rc != EA_OK) return rc;
  int *d_dist;
  float *d_plain;
  rc = dt_from_mask(p, ws, d_inv, height, width, normalize, norm_lo, norm_hi, &d_dist, &d_plain);
  if (rc != EA_OK) return rc;
  if (edges_out) HIPCHK(hipMemcpy(edges_out, d_edges, np, hipMemcpyDeviceToHost));
  if (chamfer_fix_out) HIPCHK(hipMemcpy(chamfer_fix_out, d_dist, np * 4, hipMemcpyDeviceToHost));
  if (dt_out) HIPCHK(hipMemcpy(dt_out, d_plain, np * 4, hipMemcpyDeviceToHost));
  if (!mask) { p->ws_now_kind = 2; p->ws_now_h = height; p->ws_now_w = width; }
  return EA_OK;
}

// ---- ROS flavour of the producers (ref: src/SolveEA.cpp:29-119): Canny(rgb, 150, 100, 3, true) on the 3-channel image
static int ros_thresholds(double t1, double t2, int *low, int *high) {
  // cv::Canny with L2gradient: min(t, 32767)^2, ordered
  if (t1 != t1 || t2 != t2) return fail(EA_ERR_INVALID_ARG, "Canny threshold is NaN");
  double lo = std::max(std::min(t1, t2), -1e4), hi = std::max(std::max(t1, t2), -1e4);
  lo = std::min(32767.0, lo); hi = std::min(32767.0, hi);
  if (lo > 0) lo *= lo;
  if (hi > 0) hi *= hi;
  *low = (int)std::floor(lo);
  *high = (int)std::floor(hi);
  return EA_OK;
}

// Frames as the ROS callbacks receive them -> the resolution the node works at, on the device: `halvings` times
// (depth: NaN -> 0, then) cv::resize(..., 0.5, 0.5) (src/ea.cpp:38, :56-62).  bgr / depth: host, full_h x full_w; the
// results land in d_bgr / d_depth (device, full >> halvings).  halvings = 0: a plain upload.
static int stage_scaled(ea_problem *p, const uint8_t *bgr, const float *depth, int full_h, int full_w, int halvings,
                        uint8_t *d_bgr, float *d_depth) {
  const size_t np = (size_t)full_h * full_w;
  if (halvings == 0) {
    HIPCHK(hipMemcpyAsync(d_bgr, bgr, np * 3, hipMemcpyHostToDevice, nullptr));
    if (depth) HIPCHK(hipMemcpyAsync(d_depth, depth, np * 4, hipMemcpyHostToDevice, nullptr));
    return EA_OK;
  }
  // stage: [bgr full | depth full | bgr half | depth half] (the ping-pong partner of the full-size pair)
  const size_t need = np * 3 + np * 4 + np / 4 * 3 + np / 4 * 4 + 1024;
  if (p->stage_bytes < need) {
    if (p->stage) { cached_free(p->stage); p->stage = nullptr; p->stage_bytes = 0; }
    HIPCHK(cached_malloc(reinterpret_cast<void **>(&p->stage), need, p->device));
    p->stage_bytes = need;
  }
  WsCarver st{p->stage};
  uint8_t *bgr_a = st.take<uint8_t>(np * 3), *bgr_b = nullptr;
  float *dep_a = st.take<float>(np), *dep_b = nullptr;
  bgr_b = st.take<uint8_t>(np / 4 * 3);
  dep_b = st.take<float>(np / 4);
  HIPCHK(hipMemcpyAsync(bgr_a, bgr, np * 3, hipMemcpyHostToDevice, nullptr));
  if (depth) HIPCHK(hipMemcpyAsync(dep_a, depth, np * 4, hipMemcpyHostToDevice, nullptr));
  int h = full_h, w = full_w;
  for (int k = 0; k < halvings; ++k) {
    const bool last = k == halvings - 1;
    uint8_t *bo = last ? d_bgr : bgr_b;
    float *dp = last ? d_depth : dep_b;
    HIPCHK(launch_resize_half_bgr8(bgr_a, h, w, bo, nullptr));
    if (depth) HIPCHK(launch_resize_half_f32(dep_a, h, w, dp, /*nan_to_zero=*/k == 0 ? 1 : 0, nullptr));
    std::swap(bgr_a, bgr_b); std::swap(dep_a, dep_b);
    h /= 2; w /= 2;
  }
  return EA_OK;
}

static int check_scaled_args(int height, int width, int halvings) {
  if (halvings < 0 || halvings > 8) return fail(EA_ERR_INVALID_ARG, "halvings out of range");
  if ((height % (1 << halvings)) != 0 || (width % (1 << halvings)) != 0)
    return fail(EA_ERR_INVALID_ARG, "frame extent must be divisible by 2^halvings");
  return EA_OK;
}

// SolveEA::setRefFrame (src/SolveEA.cpp:29-82): every edge pixel, depth CV_32F in metres, Z == 0 -> 1.0
extern "C" int ea_problem_set_ref_frame_ros_scaled(ea_problem *p, const uint8_t *bgr, const float *depth, int full_height,
                                                   int full_width, int halvings, double threshold1, double threshold2) {
  int rc = check_frame_args(p, bgr, full_height, full_width);
  if (rc != EA_OK) return rc;
  if (!depth) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  rc = check_scaled_args(full_height, full_width, halvings);
  if (rc != EA_OK) return rc;
  int lo, hi;
  rc = ros_thresholds(threshold1, threshold2, &lo, &hi);
  if (rc != EA_OK) return rc;
  const int height = full_height >> halvings, width = full_width >> halvings;
  if (height < 3 || width < 3) return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  HIPCHK(hipSetDevice(p->device));
  rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  const size_t np = (size_t)height * width;
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  float *d_depth = ws.take<float>(np);
  const int nblocks = (int)((np + 1023) / 1024);
  int *d_counts = ws.take<int>(nblocks + 1);
  int *d_total = d_counts + nblocks;
  rc = stage_scaled(p, bgr, depth, full_height, full_width, halvings, d_bgr, d_depth);
  if (rc != EA_OK) return rc;
  uint8_t *d_edges, *d_inv;
  rc = run_canny(ws, d_bgr, nullptr, height, width, lo, hi, &d_edges, &d_inv, nullptr, /*l2_bgr=*/1);
  if (rc != EA_OK) return rc;
  HIPCHK(launch_edge_count_scan(d_edges, nullptr, height, width, 0, d_counts, d_total, nullptr));
  int total = 0;
  HIPCHK(hipMemcpy(&total, d_total, sizeof(int), hipMemcpyDeviceToHost));
  p->version++;
  rc = reserve_points(p, total);
  if (rc != EA_OK) return rc;
  if (total > 0) {
    HIPCHK(launch_edge_scatter_ros(p->dtype, d_edges, d_depth, height, width, d_counts, p->cam.fx, p->cam.fy, p->cam.cx,
                                   p->cam.cy, p->d_x, p->d_y, p->d_z, total, nullptr));
    HIPCHK(hipDeviceSynchronize());
  }
  p->n = total;
  return EA_OK;
}

extern "C" int ea_problem_set_ref_frame_ros(ea_problem *p, const uint8_t *bgr, const float *depth, int height, int width,
                                            double threshold1, double threshold2) {
  return ea_problem_set_ref_frame_ros_scaled(p, bgr, depth, height, width, 0, threshold1, threshold2);
}

// SolveEA::setNowFrame (src/SolveEA.cpp:86-119): Canny -> 255 - edges -> distanceTransform(L2, DIST_MASK_PRECISE) ->
// normalize to [0, 255].  An image without a single edge has no defined result upstream either: EA_ERR_STATE.
static int now_frame_ros_impl(ea_problem *p, const uint8_t *bgr, int full_height, int full_width, int halvings,
                              double threshold1, double threshold2, uint8_t *edges_out, float *dt_out) {
  int rc = check_frame_args(p, bgr, full_height, full_width);
  if (rc != EA_OK) return rc;
  rc = check_scaled_args(full_height, full_width, halvings);
  if (rc != EA_OK) return rc;
  int lo, hi;
  rc = ros_thresholds(threshold1, threshold2, &lo, &hi);
  if (rc != EA_OK) return rc;
  const int height = full_height >> halvings, width = full_width >> halvings;
  if (height < 3 || width < 3) return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  HIPCHK(hipSetDevice(p->device));
  rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  const size_t np = (size_t)height * width;
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  const int nblocks = (int)((np + 1023) / 1024);
  int *d_counts = ws.take<int>(nblocks + 1);
  rc = stage_scaled(p, bgr, nullptr, full_height, full_width, halvings, d_bgr, nullptr);
  if (rc != EA_OK) return rc;
  uint8_t *d_edges, *d_inv;
  rc = run_canny(ws, d_bgr, nullptr, height, width, lo, hi, &d_edges, &d_inv, nullptr, /*l2_bgr=*/1);
  if (rc != EA_OK) return rc;
  HIPCHK(launch_edge_count_scan(d_edges, nullptr, height, width, 0, d_counts, d_counts + nblocks, nullptr));
  int total = 0;
  HIPCHK(hipMemcpy(&total, d_counts + nblocks, sizeof(int), hipMemcpyDeviceToHost));
  if (total == 0) return fail(EA_ERR_STATE, "no edge in the frame: the exact distance transform is undefined");
  int *d_dist;
  float *d_plain;
  rc = dt_from_mask(p, ws, d_inv, height, width, 1, 0.0, 255.0, &d_dist, &d_plain, /*precise=*/true);
  if (rc != EA_OK) return rc;
  if (edges_out) HIPCHK(hipMemcpy(edges_out, d_edges, np, hipMemcpyDeviceToHost));
  if (dt_out) HIPCHK(hipMemcpy(dt_out, d_plain, np * 4, hipMemcpyDeviceToHost));
  return EA_OK;
}

extern "C" int ea_problem_debug_now_frame_ros(ea_problem *p, const uint8_t *bgr, int height, int width, double threshold1,
                                              double threshold2, uint8_t *edges_out, float *dt_out) {
  return now_frame_ros_impl(p, bgr, height, width, 0, threshold1, threshold2, edges_out, dt_out);
}

extern "C" int ea_problem_set_now_frame_ros(ea_problem *p, const uint8_t *bgr, int height, int width, double threshold1,
                                            double threshold2) {
  return now_frame_ros_impl(p, bgr, height, width, 0, threshold1, threshold2, nullptr, nullptr);
}

extern "C" int ea_problem_set_now_frame_ros_scaled(ea_problem *p, const uint8_t *bgr, int full_height, int full_width,
                                                   int halvings, double threshold1, double threshold2) {
  return now_frame_ros_impl(p, bgr, full_height, full_width, halvings, threshold1, threshold2, nullptr, nullptr);
}

// The half-resolution step by itself (host in, host out) for parity checks and for callers that build pyramid levels of
// their own: kind 0 = bgr8 (height x width x 3 bytes), 1 = float32 with NaN -> 0 first (the depth callback, src/ea.cpp:56-62),
// 2 = float32 as is.  dst: (height / 2) x (width / 2) of the same element type.
extern "C" int ea_resize_half(int device, int kind, const void *src, int height, int width, void *dst) {
  if (!src || !dst) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (kind < 0 || kind > 2) return fail(EA_ERR_INVALID_ARG, "kind must be 0 (bgr8), 1 (float32, NaN -> 0) or 2 (float32)");
  if (height < 2 || width < 2 || (height & 1) || (width & 1) || (int64_t)height * width > 0x3fffffff)
    return fail(EA_ERR_INVALID_ARG, "frame extent must be even and in range");
  int rc = check_device(device);
  if (rc != EA_OK) return rc;
  HIPCHK(hipSetDevice(device));
  const size_t np = (size_t)height * width, es = kind == 0 ? 3 : 4;
  DevBuf a, b;
  HIPCHK(cached_malloc(&a.p, np * es, device));
  HIPCHK(cached_malloc(&b.p, np / 4 * es, device));
  HIPCHK(hipMemcpy(a.p, src, np * es, hipMemcpyHostToDevice));
  if (kind == 0) HIPCHK(launch_resize_half_bgr8(a.as<uint8_t>(), height, width, b.as<uint8_t>(), nullptr));
  else HIPCHK(launch_resize_half_f32(a.as<float>(), height, width, b.as<float>(), kind == 1 ? 1 : 0, nullptr));
  HIPCHK(hipMemcpy(dst, b.p, np / 4 * es, hipMemcpyDeviceToHost));
  return EA_OK;
}

extern "C" int ea_problem_set_now_frame_canny(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, int height, int width,
                                              double low_threshold, double high_threshold, int normalize, double norm_lo,
                                              double norm_hi) {
  return now_frame_canny(p, bgr, mask, height, width, low_threshold, high_threshold, normalize, norm_lo, norm_hi, nullptr,
                         nullptr, nullptr, nullptr);
}

extern "C" int ea_problem_debug_now_frame_canny(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, int height,
                                                int width, double low_threshold, double high_threshold, int normalize,
                                                double norm_lo, double norm_hi, uint8_t *edges_out,
                                                int32_t *chamfer_fix_out, float *dt_out, int *hysteresis_launches) {
  return now_frame_canny(p, bgr, mask, height, width, low_threshold, high_threshold, normalize, norm_lo, norm_hi, edges_out,
                         chamfer_fix_out, dt_out, hysteresis_launches);
}

extern "C" int ea_problem_set_now_frame(ea_problem *p, const uint8_t *bgr, int height, int width, int threshold,
                                        int median, int normalize) {
  if (!p || !bgr) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (height < 3 || width < 3 || height > 32768 || width > 32768) return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  HIPCHK(hipSetDevice(p->device));
  int rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  WsCarver ws{p->ws};
  uint8_t *d_bgr = ws.take<uint8_t>((size_t)height * width * 3);
  HIPCHK(hipMemcpyAsync(d_bgr, bgr, (size_t)height * width * 3, hipMemcpyHostToDevice, nullptr));
  rc = run_dt(p, ws, d_bgr, height, width, threshold, median, normalize, nullptr, nullptr, nullptr, nullptr);
  if (rc == EA_OK) { p->ws_now_kind = 1; p->ws_now_h = height; p->ws_now_w = width; }
  return rc;
}

// stages of the DT producer for parity checks: any output may be NULL
extern "C" int ea_problem_debug_now_frame(ea_problem *p, const uint8_t *bgr, int height, int width, int threshold,
                                          int median, int normalize, uint8_t *lap_out, uint8_t *mask_out,
                                          int32_t *chamfer_fix_out, float *dt_out) {
  if (!p || !bgr) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (height < 3 || width < 3 || height > 32768 || width > 32768) return fail(EA_ERR_INVALID_ARG, "image extent out of range");
  HIPCHK(hipSetDevice(p->device));
  int rc = ensure_ws(p, frame_ws_bytes(height, width));
  if (rc != EA_OK) return rc;
  WsCarver ws{p->ws};
  const size_t np = (size_t)height * width;
  uint8_t *d_bgr = ws.take<uint8_t>(np * 3);
  HIPCHK(hipMemcpyAsync(d_bgr, bgr, np * 3, hipMemcpyHostToDevice, nullptr));
  uint8_t *d_lap, *d_mask;
  int *d_dist;
  float *d_plain;
  rc = run_dt(p, ws, d_bgr, height, width, threshold, median, normalize, &d_lap, &d_mask, &d_dist, &d_plain);
  if (rc != EA_OK) return rc;
  if (lap_out) HIPCHK(hipMemcpy(lap_out, d_lap, np, hipMemcpyDeviceToHost));
  if (mask_out) HIPCHK(hipMemcpy(mask_out, d_mask, np, hipMemcpyDeviceToHost));
  if (chamfer_fix_out) HIPCHK(hipMemcpy(chamfer_fix_out, d_dist, np * 4, hipMemcpyDeviceToHost));
  if (dt_out) HIPCHK(hipMemcpy(dt_out, d_plain, np * 4, hipMemcpyDeviceToHost));
  return EA_OK;
}

// read back what the problem holds in HBM: points as n x 3 doubles, DT as H x W doubles ([v][u])
extern "C" int ea_problem_get_points(ea_problem *p, double *xyz, int64_t capacity) {
  if (!p || (!xyz && p->n > 0)) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (capacity < p->n) return fail(EA_ERR_INVALID_ARG, "capacity smaller than the number of points");
  if (p->n == 0) return EA_OK;
  HIPCHK(hipSetDevice(p->device));
  const size_t n = (size_t)p->n, esz = p->dtype == EA_F32 ? 4 : 8;
  std::vector<unsigned char> buf(3 * n * esz);
  HIPCHK(hipMemcpy(buf.data(), p->d_x, n * esz, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(buf.data() + n * esz, p->d_y, n * esz, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(buf.data() + 2 * n * esz, p->d_z, n * esz, hipMemcpyDeviceToHost));
  const int32_t *ord = p->order.empty() ? nullptr : p->order.data();
  for (int c = 0; c < 3; ++c)
    for (size_t i = 0; i < n; ++i)
      xyz[3 * (ord ? (size_t)ord[i] : i) + c] = p->dtype == EA_F32 ? (double)reinterpret_cast<float *>(buf.data())[c * n + i]
                                                                  : reinterpret_cast<double *>(buf.data())[c * n + i];
  return EA_OK;
}

extern "C" int ea_problem_get_dt(ea_problem *p, double *image, int *height, int *width) {
  if (!p) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (height) *height = p->H;
  if (width) *width = p->W;
  if (!image) return EA_OK;
  if (!p->d_dt) return fail(EA_ERR_STATE, "distance-transform image not set");
  HIPCHK(hipSetDevice(p->device));
  const size_t esz = p->dtype == EA_F32 ? 4 : 8;
  const size_t rows = (size_t)p->H + 2 * kImagePad;
  std::vector<unsigned char> buf((size_t)p->pitch * rows * esz);
  HIPCHK(hipMemcpy(buf.data(), p->d_dt, buf.size(), hipMemcpyDeviceToHost));
  for (int v = 0; v < p->H; ++v)
    for (int u = 0; u < p->W; ++u) {
      const size_t idx = (size_t)(v + kImagePad) * p->pitch + (u + kImagePad);
      image[(size_t)v * p->W + u] = p->dtype == EA_F32 ? (double)reinterpret_cast<float *>(buf.data())[idx]
                                                       : reinterpret_cast<double *>(buf.data())[idx];
    }
  return EA_OK;
}

#ifdef EA_STAMPS
namespace ea { hipError_t set_stamp_buffer(unsigned long long *buf); hipError_t set_lm_stamp_buffer(unsigned long long *buf); }
static unsigned long long *g_lm_stamps_dev = nullptr;
// diagnostic library only: stamps of the LM step kernel of problem 0 for the next solve(s); 64 x 8 words
extern "C" int ea_debug_lm_stamps_begin(void) {
  if (!g_lm_stamps_dev) HIPCHK(hipMalloc(&g_lm_stamps_dev, 128 * 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(g_lm_stamps_dev, 0, 128 * 8 * sizeof(unsigned long long)));
  HIPCHK(set_lm_stamp_buffer(g_lm_stamps_dev));
  return EA_OK;
}
extern "C" int ea_debug_lm_stamps_end(unsigned long long *out) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(set_lm_stamp_buffer(nullptr));
  HIPCHK(hipMemcpy(out, g_lm_stamps_dev, 128 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return EA_OK;
}
// diagnostic library only: run one fused evaluation and return the s_memtime stamps (8 per workgroup)
extern "C" int ea_debug_eval_stamps(ea_batch *b, const double *q, const double *t, unsigned long long *stamps,
                                    int64_t capacity_wgs, int64_t *n_wgs, int gx_gy[2]) {
  int rc = batch_build(b);
  if (rc != EA_OK) return rc;
  rc = batch_upload_poses(b, q, t);
  if (rc != EA_OK) return rc;
  const int gx = b->xcd_remap ? ((b->max_chunks + 7) / 8) * 8 : b->max_chunks, gy = b->nterms;
  const int64_t wgs = (int64_t)gx * gy;
  if (wgs > capacity_wgs) return fail(EA_ERR_INVALID_ARG, "stamp buffer too small");
  unsigned long long *d = nullptr;
  HIPCHK(hipMalloc(&d, wgs * 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(d, 0, wgs * 8 * sizeof(unsigned long long)));
  for (int i = 0; i < 3; ++i) { rc = batch_launch_eval(b); if (rc != EA_OK) return rc; }
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(set_stamp_buffer(d));
  rc = batch_launch_eval(b);
  if (rc != EA_OK) return rc;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(set_stamp_buffer(nullptr));
  HIPCHK(hipMemcpy(stamps, d, wgs * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  *n_wgs = wgs; gx_gy[0] = gx; gx_gy[1] = gy;
  return EA_OK;
}
#endif

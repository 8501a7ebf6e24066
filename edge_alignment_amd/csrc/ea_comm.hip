// ea_comm.hip — the multi-GPU entry points of include/ea_hip.h below Python: one communicator rank per GPU, RCCL over
// xGMI, called from librccl directly.
//
// The path shards by INDEPENDENT frame pairs (one ceres::Solve per pair, standalone_edge_align.cpp:286): every GPU solves
// its own batch with no data-path collective, and the one exchange step is the all-gather of the solved poses
// (ea_comm_gather_poses: ncclAllGather of count x 8 doubles on the batch's stream).  The other mode -- one problem
// sharded by points -- needs one all-reduce of the 32 accumulator slots per trust-region iteration
// (ea_solve_sharded_comm: ncclAllReduce enqueued by the library itself on the solve's stream, no callback, no host hop).
//
// librccl is opened with dlopen on first use: a caller that never touches ea_comm_* neither loads nor links it, and a
// process that already holds an RCCL (PyTorch-ROCm ships its own, SONAME librccl.so.1) gets THAT one -- the one bound to
// the HIP runtime this library resolved to as well when torch was loaded first.  Two HIP runtimes in one process (this
// library loaded before torch) are refused: a stream of one runtime is not an object of the other.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <link.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/ea_hip.h"
#include "ea_types.h"

// ea_capi.hip
extern "C" int ea_internal_fail(int code, const char *msg);
extern "C" void *ea_internal_batch_stream(ea_batch *b, int *device);
extern "C" int ea_internal_solve_sharded_rows(ea_problem *p, const ea_options *opt, ea_device_allreduce_fn allreduce,
                                              int (*agree)(int vals[2], void *user), void *user, double q[4], double t[3],
                                              ea_summary *summary, int *used);

namespace {

int fail(int code, const std::string &msg) { return ea_internal_fail(code, msg.c_str()); }

struct Rccl {
  void *handle = nullptr;
  std::string error;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int count_hip_runtimes() {
  int n = 0;
  dl_iterate_phdr(
      [](struct dl_phdr_info *info, size_t, void *data) {
        const char *name = info->dlpi_name ? info->dlpi_name : "";
        const char *base = std::strrchr(name, '/');
        base = base ? base + 1 : name;
        if (std::strncmp(base, "libamdhip64.so", 14) == 0) ++*static_cast<int *>(data);
        return 0;
      },
      &n);
  return n;
}

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) { r.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
#define EA_RCCL_SYM(name)                                                                        \
  r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.handle, "nccl" #name));                    \
  if (!r.name && r.error.empty()) r.error = "librccl lacks nccl" #name;
    EA_RCCL_SYM(GetUniqueId) EA_RCCL_SYM(CommInitRank) EA_RCCL_SYM(CommInitAll) EA_RCCL_SYM(CommDestroy)
    EA_RCCL_SYM(AllGather) EA_RCCL_SYM(AllReduce) EA_RCCL_SYM(GetErrorString)
#undef EA_RCCL_SYM
  });
  return r;
}

int rccl_ready() {
  Rccl &r = rccl();
  if (!r.error.empty()) return fail(EA_ERR_STATE, r.error);
  const int copies = count_hip_runtimes();
  if (copies > 1)
    return fail(EA_ERR_STATE, "two HIP runtimes are mapped in this process (libea_hip.so was loaded before PyTorch-ROCm brought its own): "
                              "a stream of one is not an object of the other -- import torch first");
  return EA_OK;
}

// RCCL announces itself on STDOUT when a communicator is created ("RCCL version : ..." and four more lines on rank 0).
// The stdout of a process that uses this library belongs to its caller (bench.py's contract is ONE JSON line there):
// while a communicator is being created, file descriptor 1 points at stderr.
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() {
    std::fflush(stdout);
    saved = dup(1);
    if (saved >= 0 && dup2(2, 1) < 0) { close(saved); saved = -1; }
  }
  ~StdoutToStderr() {
    if (saved < 0) return;
    std::fflush(stdout);
    (void)dup2(saved, 1);
    close(saved);
  }
};

#define NCCLCHK(expr)                                                                                         \
  do {                                                                                                        \
    const ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) return fail(EA_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_));    \
  } while (0)
#define HIPCHK(expr)                                                                                          \
  do {                                                                                                        \
    const hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(e_ == hipErrorNoDevice ? EA_ERR_NO_DEVICE : EA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace

struct ea_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;        // for gathers that are not ordered behind a batch
  // pose gather: send = cap x 8 doubles, recv = nranks x cap x 8 (device), the same in pinned host memory
  int cap = 0;
  double *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;
  double *d_sums = nullptr;            // the 32 accumulator slots of ea_solve_sharded_comm
  int *d_agree = nullptr;              // two ints the ranks take to their maximum before a sharded solve
  hipStream_t last_solve_stream = nullptr;  // collectives of the last sharded solve may still be draining on it (waited for in destroy)
  int64_t allreduces = 0, allgathers = 0, row_solves = 0;
  bool spoke = false;                  // the first collective has run (some RCCL builds announce themselves there, not at init)
};

extern "C" int ea_hip_runtime_copies(void) { return count_hip_runtimes(); }

extern "C" int ea_comm_get_unique_id(unsigned char id[EA_COMM_ID_BYTES]) {
  static_assert(EA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  if (!id) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  int rc = rccl_ready();
  if (rc != EA_OK) return rc;
  ncclUniqueId u;
  {
    StdoutToStderr quiet;
    NCCLCHK(rccl().GetUniqueId(&u));
  }
  std::memcpy(id, u.internal, EA_COMM_ID_BYTES);
  return EA_OK;
}

static int comm_finish(ea_comm *c) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_sums), ea::kAccSlots * sizeof(double)));
  HIPCHK(hipMemset(c->d_sums, 0, ea::kAccSlots * sizeof(double)));
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_agree), 2 * sizeof(int)));
  return EA_OK;
}

extern "C" void ea_comm_destroy(ea_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->last_solve_stream) (void)hipStreamSynchronize(c->last_solve_stream);
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  (void)hipFree(c->d_send); (void)hipFree(c->d_recv); (void)hipFree(c->d_sums); (void)hipFree(c->d_agree);
  (void)hipHostFree(c->h_send); (void)hipHostFree(c->h_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int ea_comm_create(ea_comm **out, const unsigned char id[EA_COMM_ID_BYTES], int nranks, int rank, int device) {
  if (!out || !id) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(EA_ERR_INVALID_ARG, "need 0 <= rank < nranks");
  int rc = rccl_ready();
  if (rc != EA_OK) return rc;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(EA_ERR_INVALID_ARG, "device index out of range");
  ea_comm *c = new (std::nothrow) ea_comm;
  if (!c) return fail(EA_ERR_ALLOC, "out of host memory");
  c->nranks = nranks; c->rank = rank; c->device = device;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { delete c; return fail(EA_ERR_HIP, hipGetErrorString(e)); }
  ncclUniqueId u;
  std::memcpy(u.internal, id, EA_COMM_ID_BYTES);
  ncclResult_t r;
  {
    StdoutToStderr quiet;
    r = rccl().CommInitRank(&c->comm, nranks, u, rank);
  }
  if (r != ncclSuccess) { c->comm = nullptr; ea_comm_destroy(c); return fail(EA_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r)); }
  rc = comm_finish(c);
  if (rc != EA_OK) { ea_comm_destroy(c); return rc; }
  *out = c;
  return EA_OK;
}

// one process driving several GPUs (host threads, one per device): all communicators of the node in one call
extern "C" int ea_comm_create_all(ea_comm **out, const int *devices, int ndev) {
  if (!out || ndev < 1) return fail(EA_ERR_INVALID_ARG, "bad argument");
  int rc = rccl_ready();
  if (rc != EA_OK) return rc;
  int have = 0;
  HIPCHK(hipGetDeviceCount(&have));
  std::vector<int> devs((size_t)ndev);
  for (int i = 0; i < ndev; ++i) {
    devs[(size_t)i] = devices ? devices[i] : i;
    if (devs[(size_t)i] < 0 || devs[(size_t)i] >= have) return fail(EA_ERR_INVALID_ARG, "device index out of range");
    for (int j = 0; j < i; ++j)
      if (devs[(size_t)j] == devs[(size_t)i]) return fail(EA_ERR_INVALID_ARG, "a device may appear once");
  }
  std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
  {
    StdoutToStderr quiet;
    const ncclResult_t r = rccl().CommInitAll(comms.data(), ndev, devs.data());
    if (r != ncclSuccess) return fail(EA_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(r));
  }
  for (int i = 0; i < ndev; ++i) out[i] = nullptr;
  for (int i = 0; i < ndev; ++i) {
    ea_comm *c = new (std::nothrow) ea_comm;
    if (!c) { rc = fail(EA_ERR_ALLOC, "out of host memory"); break; }
    c->comm = comms[(size_t)i]; comms[(size_t)i] = nullptr;
    c->nranks = ndev; c->rank = i; c->device = devs[(size_t)i];
    out[i] = c;
    if ((rc = comm_finish(c)) != EA_OK) break;
  }
  if (rc != EA_OK) {
    for (int i = 0; i < ndev; ++i) {
      if (out[i]) ea_comm_destroy(out[i]);
      else if (comms[(size_t)i]) (void)rccl().CommDestroy(comms[(size_t)i]);
      out[i] = nullptr;
    }
  }
  return rc;
}

extern "C" int ea_comm_rank(const ea_comm *c) { return c ? c->rank : -1; }
extern "C" int ea_comm_size(const ea_comm *c) { return c ? c->nranks : 0; }

static int gather_reserve(ea_comm *c, int count) {
  if (count <= c->cap) return EA_OK;
  (void)hipFree(c->d_send); (void)hipFree(c->d_recv); (void)hipHostFree(c->h_send); (void)hipHostFree(c->h_recv);
  c->d_send = c->d_recv = c->h_send = c->h_recv = nullptr;
  c->cap = 0;
  const size_t one = (size_t)count * 8 * sizeof(double), all = one * (size_t)c->nranks;
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_send), one));
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_recv), all));
  HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_send), one, hipHostMallocDefault));
  HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_recv), all, hipHostMallocDefault));
  c->cap = count;
  return EA_OK;
}

// THE collective of the batch mode: every rank contributes the `count` poses it solved (q: count x 4, t: count x 3,
// status: count ints, e.g. ea_summary.termination; NULL = zeros) and receives all nranks x count of them in rank order.
// One ncclAllGather of count x 8 doubles, enqueued on `after`'s stream (the batch that produced the poses; NULL: the
// communicator's own stream) between the two staging copies, one synchronisation.  `count` must be the same on every rank.
extern "C" int ea_comm_gather_poses(ea_comm *c, ea_batch *after, const double *q, const double *t, const int *status, int count,
                                    double *all_q, double *all_t, int *all_status) {
  if (!c || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  if (count < 1 || count > (1 << 24)) return fail(EA_ERR_INVALID_ARG, "count out of range");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  if (after) {
    int dev = -1;
    st = static_cast<hipStream_t>(ea_internal_batch_stream(after, &dev));
    if (dev != c->device) return fail(EA_ERR_INVALID_ARG, "the batch lives on another device than the communicator");
  }
  int rc = gather_reserve(c, count);
  if (rc != EA_OK) return rc;
  for (int i = 0; i < count; ++i) {
    double *d = c->h_send + 8 * (size_t)i;
    d[0] = q[4 * i]; d[1] = q[4 * i + 1]; d[2] = q[4 * i + 2]; d[3] = q[4 * i + 3];
    d[4] = t[3 * i]; d[5] = t[3 * i + 1]; d[6] = t[3 * i + 2];
    d[7] = status ? (double)status[i] : 0.0;
  }
  const size_t one = (size_t)count * 8;
  HIPCHK(hipMemcpyAsync(c->d_send, c->h_send, one * sizeof(double), hipMemcpyHostToDevice, st));
  if (!c->spoke) {
    StdoutToStderr quiet;
    c->spoke = true;
    NCCLCHK(rccl().AllGather(c->d_send, c->d_recv, one, ncclDouble, c->comm, st));
  } else {
    NCCLCHK(rccl().AllGather(c->d_send, c->d_recv, one, ncclDouble, c->comm, st));
  }
  HIPCHK(hipMemcpyAsync(c->h_recv, c->d_recv, one * (size_t)c->nranks * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  c->allgathers += 1;
  const size_t n = (size_t)count * (size_t)c->nranks;
  for (size_t i = 0; i < n; ++i) {
    const double *s = c->h_recv + 8 * i;
    if (all_q) { all_q[4 * i] = s[0]; all_q[4 * i + 1] = s[1]; all_q[4 * i + 2] = s[2]; all_q[4 * i + 3] = s[3]; }
    if (all_t) { all_t[3 * i] = s[4]; all_t[3 * i + 1] = s[5]; all_t[3 * i + 2] = s[6]; }
    if (all_status) all_status[i] = (int)s[7];
  }
  return EA_OK;
}

// One problem sharded by points with the per-iteration exchange issued by the library: evaluation ->
// fold -> ncclAllReduce(32 doubles, sum) -> step kernel, all enqueued on the solve's stream; nothing leaves the device
// and no callback runs between iterations.  Every rank calls it with its shard, the same options and start pose.
extern "C" int ea_solve_sharded_comm(ea_problem *p, const ea_options *opt, ea_comm *c, double q[4], double t[3], ea_summary *summary) {
  if (!p || !c || !q || !t) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  auto enqueue = [](void *buf, int count, void *stream, void *user) -> int {
    ea_comm *cc = static_cast<ea_comm *>(user);
    cc->last_solve_stream = static_cast<hipStream_t>(stream);
    ncclResult_t r;
    if (!cc->spoke) {
      StdoutToStderr quiet;
      cc->spoke = true;
      r = rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, cc->comm, static_cast<hipStream_t>(stream));
    } else {
      r = rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, cc->comm, static_cast<hipStream_t>(stream));
    }
    if (r != ncclSuccess) {
      ea_internal_fail(EA_ERR_HIP, (std::string("ncclAllReduce: ") + rccl().GetErrorString(r)).c_str());
      return 1;
    }
    cc->allreduces += 1;
    return 0;
  };
  // First choice: one launch per iteration with the partial ROWS all-reduced in place (ea_internal_solve_sharded_rows) -- every
  // rank's shard has to qualify, which the ranks settle with one small MAX all-reduce up front.  EA_SHARDED_ROWS=0 in the
  // environment keeps the (evaluate, fold, all-reduce, step) form (A/B, tests).
  const char *env = std::getenv("EA_SHARDED_ROWS");
  if (!(env && env[0] == '0')) {
    auto agree = [](int vals[2], void *user) -> int {
      ea_comm *cc = static_cast<ea_comm *>(user);
      if (hipMemcpyAsync(cc->d_agree, vals, 2 * sizeof(int), hipMemcpyHostToDevice, cc->stream) != hipSuccess) return 1;
      ncclResult_t r;
      {
        StdoutToStderr quiet;  // (may be the communicator's first collective)
        cc->spoke = true;
        r = rccl().AllReduce(cc->d_agree, cc->d_agree, 2, ncclInt, ncclMax, cc->comm, cc->stream);
      }
      if (r != ncclSuccess) return 1;
      if (hipMemcpyAsync(vals, cc->d_agree, 2 * sizeof(int), hipMemcpyDeviceToHost, cc->stream) != hipSuccess) return 1;
      return hipStreamSynchronize(cc->stream) == hipSuccess ? 0 : 1;
    };
    int used = 0;
    int rc = ea_internal_solve_sharded_rows(p, opt, enqueue, agree, c, q, t, summary, &used);
    if (rc != EA_OK) return rc;
    if (used) { c->row_solves += 1; return EA_OK; }
  }
  return ea_solve_sharded_device(p, opt, enqueue, c, c->d_sums, q, t, summary);
}

extern "C" int ea_comm_get_info(const ea_comm *c, const char *key, int64_t *value) {
  if (!c || !key || !value) return fail(EA_ERR_INVALID_ARG, "NULL argument");
  const std::string k(key);
  if (k == "allreduces") *value = c->allreduces;
  else if (k == "allgathers") *value = c->allgathers;
  else if (k == "row_solves") *value = c->row_solves;  // sharded solves that exchanged partial rows (one launch per iteration)
  else if (k == "device") *value = c->device;
  else return fail(EA_ERR_INVALID_ARG, "unknown info key: " + k);
  return EA_OK;
}

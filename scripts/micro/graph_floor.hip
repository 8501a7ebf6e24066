// What a kernel node of a replayed hipGraph costs on gfx950 when the kernel does (almost) nothing: the floor under the
// 3.15 us the C2 step takes (196 workgroups x 256 threads + the riding fold's workgroup, profiles/r02_bench_c2.json).
// Kernels: empty; one dependent global load -> store per lane (one memory round trip); two dependent round trips
// (load an index, load through it, store) -- the evaluation kernel's chain is descriptor -> points -> stencil rows -> row store.
// Each as a graph of 2000 nodes in one stream, replayed 5 times, event pair around the replay, best.
//   hipcc --offload-arch=gfx950 -O3 -o graph_floor graph_floor.hip && ./graph_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(const int *, const double *, double *) {}
__global__ void k_one_trip(const int *, const double *in, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] * 2.0;
}
__global__ void k_two_trips(const int *idx, const double *in, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[idx[i]] * 2.0;
}
__global__ void k_three_trips(const int *idx, const double *in, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[idx[idx[i]]] * 2.0;
}
typedef void (*kern_t)(const int *, const double *, double *);
static int run(const char *name, kern_t k, int blocks, int threads, const int *idx, const double *in, double *out) {
  const int K = 2000;
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, s, idx, in, out);
  hipGraph_t g; CHECK(hipStreamEndCapture(s, &g));
  hipGraphExec_t ge; CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::printf("%-34s %4d workgroups x %4d threads: %6.3f us per node\n", name, blocks, threads, best * 1e3f / K);
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
  return 0;
}
int main() {
  const int n = 1024 * 1024;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919) % n);
  int *idx; double *in, *out;
  CHECK(hipMalloc(&idx, n * sizeof(int))); CHECK(hipMalloc(&in, n * sizeof(double))); CHECK(hipMalloc(&out, n * sizeof(double)));
  CHECK(hipMemcpy(idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice)); CHECK(hipMemset(in, 0, n * sizeof(double)));
  const int shapes[][2] = {{1, 64}, {196, 256}, {197, 256}, {256, 256}, {49, 1024}, {1024, 256}};
  for (auto &sh : shapes) {
    if (run("empty", k_empty, sh[0], sh[1], idx, in, out)) return 1;
    if (run("one memory round trip", k_one_trip, sh[0], sh[1], idx, in, out)) return 1;
    if (run("two dependent round trips", k_two_trips, sh[0], sh[1], idx, in, out)) return 1;
    if (run("three dependent round trips", k_three_trips, sh[0], sh[1], idx, in, out)) return 1;
  }
  return 0;
}

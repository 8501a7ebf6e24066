// Cost of COLD instruction fetch on gfx950: the same number of dependent-free v_fma_f32 per lane executed as straight-line
// code (N x 8 bytes of instructions, every cache line touched once) and as a loop over a 32-instruction body (256 bytes of
// code), launched back to back.  One wavefront per workgroup; 1 workgroup or 256 (one per CU).
//   hipcc --offload-arch=gfx950 -O3 -o ifetch ifetch.hip && ./ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#define F(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(i) & 7]) : "v"(b));
#define R8(i) F(i) F(i + 1) F(i + 2) F(i + 3) F(i + 4) F(i + 5) F(i + 6) F(i + 7)
#define R32(i) R8(i) R8(i + 8) R8(i + 16) R8(i + 24)
#define R128(i) R32(i) R32(i) R32(i) R32(i)
#define R512(i) R128(i) R128(i) R128(i) R128(i)
#define R2048(i) R512(i) R512(i) R512(i) R512(i)
template <int N, int TAG = 0>
__global__ void straight(float *out) {
  float a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  float b = threadIdx.x * 0.5f + 1.f;
  if (N >= 512) { R512(0) }
  if (N >= 2048) { R512(0) R512(0) R512(0) }
  if (N >= 8192) { R2048(0) R2048(0) R2048(0) }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void looped(float *out, int n) {
  float a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  float b = threadIdx.x * 0.5f + 1.f;
  for (int it = 0; it < n / 32; ++it) { R32(0) }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename L>
static float time_b2b(L launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
int main() {
  float *out; hipMalloc(&out, 1 << 20);
  for (int grid : {1, 256}) {
    printf("grid = %d workgroup(s) of one wavefront; us per launch, back to back\n", grid);
    float l0 = time_b2b([&] { hipLaunchKernelGGL(looped, dim3(grid), dim3(64), 0, 0, out, 0); }, 500);
    printf("  empty body                          %6.2f us\n", l0);
    float s512 = time_b2b([&] { hipLaunchKernelGGL(straight<512>, dim3(grid), dim3(64), 0, 0, out); }, 500);
    float l512 = time_b2b([&] { hipLaunchKernelGGL(looped, dim3(grid), dim3(64), 0, 0, out, 512); }, 500);
    printf("  512 FMAs   straight (4 KB)  %6.2f us   looped %6.2f us   cold-fetch cost %5.2f us\n", s512, l512, s512 - l512);
    float s2k = time_b2b([&] { hipLaunchKernelGGL(straight<2048>, dim3(grid), dim3(64), 0, 0, out); }, 500);
    float l2k = time_b2b([&] { hipLaunchKernelGGL(looped, dim3(grid), dim3(64), 0, 0, out, 2048); }, 500);
    printf("  2048 FMAs  straight (16 KB) %6.2f us   looped %6.2f us   cold-fetch cost %5.2f us\n", s2k, l2k, s2k - l2k);
    float s8k = time_b2b([&] { hipLaunchKernelGGL(straight<8192>, dim3(grid), dim3(64), 0, 0, out); }, 300);
    float l8k = time_b2b([&] { hipLaunchKernelGGL(looped, dim3(grid), dim3(64), 0, 0, out, 8192); }, 300);
    printf("  8192 FMAs  straight (64 KB) %6.2f us   looped %6.2f us   cold-fetch cost %5.2f us\n", s8k, l8k, s8k - l8k);
  }
  // kernel switches: the same two kernels launched AABB... (each behind itself) and ABAB... (each behind the other)
  for (int grid : {1, 256}) {
    auto A = [&] { hipLaunchKernelGGL((straight<2048, 0>), dim3(grid), dim3(64), 0, 0, out); };
    auto B = [&] { hipLaunchKernelGGL((straight<2048, 1>), dim3(grid), dim3(64), 0, 0, out); };
    auto C = [&] { hipLaunchKernelGGL((straight<8192, 1>), dim3(grid), dim3(64), 0, 0, out); };
    float a = time_b2b(A, 400), b = time_b2b(B, 400), c = time_b2b(C, 300);
    float ab = time_b2b([&] { A(); B(); }, 300);
    float ac = time_b2b([&] { A(); C(); }, 200);
    printf("grid %3d: A (16 KB) %5.2f  B (16 KB) %5.2f  C (64 KB) %5.2f us alone;  A,B alternating %5.2f us per pair (sum %5.2f);  A,C alternating %5.2f (sum %5.2f)\n",
           grid, a, b, c, ab, a + b, ac, a + c);
  }
  return 0;
}

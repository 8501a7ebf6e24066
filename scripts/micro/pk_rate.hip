// Issue-rate micro-benchmark: v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64 on gfx950 (cycles per wave64 instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float F2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, int iters, unsigned long long *cycles) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  F2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const float m = 1.0000001f, c = 1e-9f;
  const F2 m2 = {m, m}, c2 = {c, c};
  const double md = 1.0000001, cd = 1e-9;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c); }
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { p0 = __builtin_elementwise_fma(p0, m2, c2); p1 = __builtin_elementwise_fma(p1, m2, c2); p2 = __builtin_elementwise_fma(p2, m2, c2); p3 = __builtin_elementwise_fma(p3, m2, c2);
        p4 = __builtin_elementwise_fma(p4, m2, c2); p5 = __builtin_elementwise_fma(p5, m2, c2); p6 = __builtin_elementwise_fma(p6, m2, c2); p7 = __builtin_elementwise_fma(p7, m2, c2); }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) { d0 = __builtin_fma(d0, md, cd); d1 = __builtin_fma(d1, md, cd); d2 = __builtin_fma(d2, md, cd); d3 = __builtin_fma(d3, md, cd);
        d4 = __builtin_fma(d4, md, cd); d5 = __builtin_fma(d5, md, cd); d6 = __builtin_fma(d6, md, cd); d7 = __builtin_fma(d7, md, cd); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const int iters = 2000;
  const char *names[3] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64"};
  for (int waves = 1; waves <= 4; waves *= 2)
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      dim3 grid(256 * 4), block(64 * waves);  // `waves` wavefronts per SIMD when 4 workgroups share a CU
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, cyc);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, cyc);
        else hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double instr = (double)iters * 64;
      printf("%-14s %d wave(s)/SIMD: %.2f counter ticks / instr (one wave), kernel %.3f ms -> %.2f ns per instr per SIMD-resident wave set\n",
             names[mode], waves, (double)h / instr, ms, ms * 1e6 / (instr * waves));
    }
  return 0;
}

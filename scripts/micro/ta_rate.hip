// Vector-memory issue cost on gfx950: ns (and 2.4 GHz cycles) per wave64 load instruction per CU for the access shapes
// the DT stencil can take -- 4 / 8 / 16 bytes per lane, coalesced or strided, 16-byte aligned or not -- from a region
// that stays in the CU's L1 (16 KB) or in the XCD's L2 (2 MB), and the LDS equivalents.
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W> struct __attribute__((packed, aligned(4))) Chunk { float v[W / 4]; };
template <int W>
__global__ __launch_bounds__(256) void k_global(const char *base, int lane_stride, int misalign, int iter_stride,
                                                unsigned mask, int iters, float *out) {
  const unsigned off = threadIdx.x * lane_stride + misalign + (blockIdx.x & 7) * 4096u;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned o = (off + (unsigned)(i * 8 + u) * (unsigned)iter_stride) & mask;
      const Chunk<W> c = *reinterpret_cast<const Chunk<W> *>(base + o);
#pragma unroll
      for (int e = 0; e < W / 4; ++e) acc += c.v[e];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int W>
__global__ __launch_bounds__(256) void k_lds(int lane_stride, int misalign, int iter_stride, unsigned mask, int iters,
                                             float *out) {
  __shared__ __attribute__((aligned(16))) char s[32768 + 64];
  for (int i = threadIdx.x; i < (32768 + 64) / 4; i += 256) reinterpret_cast<float *>(s)[i] = (float)i;
  __syncthreads();
  const unsigned off = threadIdx.x * lane_stride + misalign;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned o = (off + (unsigned)(i * 8 + u) * (unsigned)iter_stride) & mask;
      const Chunk<W> c = *reinterpret_cast<const Chunk<W> *>(s + o);
#pragma unroll
      for (int e = 0; e < W / 4; ++e) acc += c.v[e];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const size_t region = (4u << 20) + 65536;
  char *buf; float *out;
  hipMalloc(&buf, region); hipMemset(buf, 0, region); hipMalloc(&out, 4096 * 256 * 4);
  const int grid = 2048, iters = 64;
  struct Case { const char *name; int w, stride, mis, iter_stride; unsigned mask; int lds; };
  const unsigned L1 = (16u << 10) - 1, L2 = (2u << 20) - 1;
  std::vector<Case> cases = {
      {"dword   stride  4 (coalesced)      L1", 4, 4, 0, 1024, L1, 0},
      {"dword   stride 24                  L1", 4, 24, 0, 2560, L1, 0},
      {"dwordx2 stride  8 (coalesced)      L1", 8, 8, 0, 2048, L1, 0},
      {"dwordx2 stride 24                  L1", 8, 24, 0, 2560, L1, 0},
      {"dwordx4 stride 16 aligned (coal.)  L1", 16, 16, 0, 4096, L1, 0},
      {"dwordx4 stride 16 +4 (unaligned)   L1", 16, 16, 4, 4096, L1, 0},
      {"dwordx4 stride 24 +4 (fp32 rows)   L1", 16, 24, 4, 2560, L1, 0},
      {"dwordx4 stride 32 aligned          L1", 16, 32, 0, 2560, L1, 0},
      {"dwordx4 stride 48 +8 (fp64 rows)   L1", 16, 48, 8, 2560, L1, 0},
      {"dwordx4 stride 16 aligned (coal.)  L2", 16, 16, 0, 69632, L2, 0},
      {"dwordx4 stride 24 +4 (fp32 rows)   L2", 16, 24, 4, 68612, L2, 0},
      {"dwordx4 stride 48 +8 (fp64 rows)   L2", 16, 48, 8, 68616, L2, 0},
      {"dword   stride 24                  L2", 4, 24, 0, 68612, L2, 0},
      {"dwordx4 stride 272 +4 (scattered)  L2", 16, 272, 4, 68612, L2, 0},
      {"ds b32  stride  4                  LDS", 4, 4, 0, 1024, 32767, 1},
      {"ds b32  stride 24                  LDS", 4, 24, 0, 2560, 32767, 1},
      {"ds b64  stride 24                  LDS", 8, 24, 0, 2560, 32767, 1},
      {"ds b128 stride 16 aligned          LDS", 16, 16, 0, 4096, 32767, 1},
      {"ds 16 B stride 24 +4 (fp32 rows)   LDS", 16, 24, 4, 2560, 32767, 1},
      {"ds 16 B stride 48 +8 (fp64 rows)   LDS", 16, 48, 8, 2560, 32767, 1},
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Case &c : cases) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (c.lds) {
        if (c.w == 4) hipLaunchKernelGGL(k_lds<4>, dim3(grid), dim3(256), 0, 0, c.stride, c.mis, c.iter_stride, c.mask, iters, out);
        else if (c.w == 8) hipLaunchKernelGGL(k_lds<8>, dim3(grid), dim3(256), 0, 0, c.stride, c.mis, c.iter_stride, c.mask, iters, out);
        else hipLaunchKernelGGL(k_lds<16>, dim3(grid), dim3(256), 0, 0, c.stride, c.mis, c.iter_stride, c.mask, iters, out);
      } else {
        if (c.w == 4) hipLaunchKernelGGL(k_global<4>, dim3(grid), dim3(256), 0, 0, buf, c.stride, c.mis, c.iter_stride, c.mask, iters, out);
        else if (c.w == 8) hipLaunchKernelGGL(k_global<8>, dim3(grid), dim3(256), 0, 0, buf, c.stride, c.mis, c.iter_stride, c.mask, iters, out);
        else hipLaunchKernelGGL(k_global<16>, dim3(grid), dim3(256), 0, 0, buf, c.stride, c.mis, c.iter_stride, c.mask, iters, out);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double wave_loads_per_cu = (double)grid * 4 * iters * 8 / 256.0;
    const double ns = ms * 1e6 / wave_loads_per_cu;
    printf("%-40s %7.3f ms  %6.1f ns / wave-load / CU  (%5.0f cycles at 2.4 GHz)  %6.1f GB/s per CU useful\n", c.name, ms, ns,
           ns * 2.4, 64.0 * c.w / ns);
  }
  return 0;
}

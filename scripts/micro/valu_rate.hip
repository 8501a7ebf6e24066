// VALU issue cost per opcode class on gfx950: ns per wave64 instruction per SIMD with 1 / 4 / 8 waves per SIMD.
// Every test is a loop of 32 inline-asm instructions over 8 independent register chains.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define KERNEL(NAME, ASM, CONSTRAINT_EXTRA)                                                              \
  __global__ void NAME(float *out, int iters, float m, float c) {                                         \
    float a[8];                                                                                           \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                   \
    float b = threadIdx.x * 0.5f + 1.0f;                                                                  \
    for (int it = 0; it < iters; ++it) {                                                                  \
      REP32(ASM)                                                                                          \
    }                                                                                                     \
    float s = 0;                                                                                          \
    for (int i = 0; i < 8; ++i) s += a[i];                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
  }
#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(b));
#define A_FMA_S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(b));
#define A_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(b));
#define A_FMAC_S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(b));
#define A_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MUL_S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(m));
#define A_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define A_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define A_FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x3fc00000, %1" : "+v"(a[i]) : "v"(b));
#define A_FMA_LIT(i) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a[i]) : "v"(b));
#define A_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define A_CVT(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define A_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define A_MIX(i) asm volatile("v_mul_f32 %0, %0, %1\n v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define A_DEP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(b));
KERNEL(k_fma, A_FMA, ) KERNEL(k_fma_s, A_FMA_S, ) KERNEL(k_fmac, A_FMAC, ) KERNEL(k_fmac_s, A_FMAC_S, ) KERNEL(k_mul, A_MUL, )
KERNEL(k_mul_s, A_MUL_S, ) KERNEL(k_add, A_ADD, ) KERNEL(k_mov, A_MOV, ) KERNEL(k_cnd, A_CND, ) KERNEL(k_fmamk, A_FMAMK, )
KERNEL(k_fma_lit, A_FMA_LIT, ) KERNEL(k_max, A_MAX, ) KERNEL(k_floor, A_FLOOR, ) KERNEL(k_cvt, A_CVT, ) KERNEL(k_rcp, A_RCP, )
KERNEL(k_dpp, A_DPP, ) KERNEL(k_add3, A_ADD3, ) KERNEL(k_mix, A_MIX, ) KERNEL(k_dep, A_DEP, )
// 64-bit forms
__global__ void k_f64(float *out, int iters, float m, float c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  double b = threadIdx.x * 0.5 + 1.0;
  for (int it = 0; it < iters; ++it) {
#define A_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
    REP32(A_F64)
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_mul64(float *out, int iters, float m, float c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  double b = threadIdx.x * 0.5 + 1.0;
  for (int it = 0; it < iters; ++it) {
#define A_M64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    REP32(A_M64)
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_add64(float *out, int iters, float m, float c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  double b = threadIdx.x * 0.5 + 1.0;
  for (int it = 0; it < iters; ++it) {
#define A_A64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    REP32(A_A64)
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_lshladd64(float *out, int iters, float m, float c) {
  unsigned long long a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  unsigned long long b = threadIdx.x * 3 + 1;
  for (int it = 0; it < iters; ++it) {
#define A_L64(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
    REP32(A_L64)
  }
  unsigned long long s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_swap(float *out, int iters, float m, float c) {
  float a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#define A_SW(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
    REP32(A_SW)
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef void (*kern_t)(float *, int, float, float);
int main() {
  float *out; hipMalloc(&out, 1 << 24);
  struct T { const char *name; kern_t k; int per_iter; };
  T tests[] = {{"v_fma_f32 (VOP3, 3 VGPR)", k_fma, 32}, {"v_fma_f32 (SGPR operand)", k_fma_s, 32}, {"v_fmac_f32 (VOP2)", k_fmac, 32},
               {"v_fmac_f32 (SGPR operand)", k_fmac_s, 32}, {"v_mul_f32", k_mul, 32}, {"v_mul_f32 (SGPR operand)", k_mul_s, 32},
               {"v_add_f32", k_add, 32}, {"v_mov_b32", k_mov, 32}, {"v_cndmask_b32", k_cnd, 32}, {"v_fmamk_f32 (literal)", k_fmamk, 32},
               {"v_fma_f32 (inline const)", k_fma_lit, 32}, {"v_max_f32", k_max, 32}, {"v_floor_f32", k_floor, 32},
               {"v_cvt_i32_f32", k_cvt, 32}, {"v_rcp_f32", k_rcp, 32}, {"v_add_f32_dpp row_mirror", k_dpp, 32},
               {"v_add3_u32", k_add3, 32}, {"v_mul_f32 + v_fmac_f32 pairs", k_mix, 64}, {"v_fma_f32 one dependent chain", k_dep, 32},
               {"v_fma_f64", k_f64, 32}, {"v_mul_f64", k_mul64, 32}, {"v_add_f64", k_add64, 32}, {"v_lshl_add_u64", k_lshladd64, 32},
               {"v_permlane32_swap_b32", k_swap, 32}};
  const int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-34s %12s %12s %12s   (ns per wave64 instruction per SIMD)\n", "", "1 wave/SIMD", "4 waves/SIMD", "8 waves/SIMD");
  for (const T &t : tests) {
    printf("%-34s", t.name);
    for (int waves : {1, 4, 8}) {
      // 256-thread workgroups (one wave per SIMD each); `waves` workgroups per CU
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(256 * waves), dim3(256), 0, 0, out, iters, 1.0000001f, 1e-9f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf(" %12.2f", ms * 1e6 / ((double)iters * t.per_iter * waves));
    }
    printf("\n");
  }
  return 0;
}

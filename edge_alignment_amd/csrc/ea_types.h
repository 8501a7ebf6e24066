// ea_types.h — structures shared by the host driver and the gfx950 kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EA_HD __host__ __device__
#else
#define EA_HD
#endif

namespace ea {

// accumulator slots of one fused evaluation (all fp64)
//   0..20  JtJ upper triangle, row-major (a <= b)
//  21..26  Jtr
//  27      cost = 1/2 sum rho(r^2)
//  28      number of blocks whose functor returned false
//  29..31  spare (zero)
constexpr int kAccJtJ = 0;
constexpr int kAccJtr = 21;
constexpr int kAccCost = 27;
constexpr int kAccInvalid = 28;
constexpr int kAccSlots = 32;

constexpr int kImagePad = 3;      // replicated border texels on every side of the DT copy in HBM
constexpr int kBlockThreads = 256;

// One frame pair as the kernels see it.  Arrays are in the problem dtype T (float/double).
// Scalars are kept in both precisions so the fp32 kernels never convert uniform values per point.
struct ProblemDesc {
  const void *x, *y, *z;  // SoA edge points (frame A), n each
  const void *dt;         // padded DT image: (H + 2*pad) rows x pitch, texel (v,u) at [(v+pad)*pitch + u+pad]
  int32_t n;
  int32_t W, H;           // u extent (Grid2D rows), v extent (Grid2D cols)
  int32_t pitch;          // elements per padded row
  double fx, fy, cx, cy;
  double loss_a, z_guard, z_eps;
  double loss_inv_b;             // 1 / loss_a^2 (uniform: formed once on the host, not per lane)
  float fxf, fyf, cxf, cyf;
  float loss_af, z_guardf, z_epsf;
  float loss_inv_bf;
  int32_t loss_kind;
  int32_t rot_transposed;
  int32_t tile_begin, tile_end;  // this term's rows in the partial-sum array (one per workgroup)
  int32_t group;                 // which pose / LM state this residual family belongs to
  // residual variants of standalone/utils.h:102-421 (variant != 0 selects the variant kernel)
  int32_t variant;               // bit 0: Brown-Conrady distortion, bit 1: second camera of a rigid rig
  int64_t row_begin;             // this term's first row in the batch's materialised outputs (ea_batch_eval_rows_device)
  const void *dt32;              // fp64 problems: the same padded image stored as float32 (same pitch in texels) when every
                                 // value is exactly float-representable -- true of everything the reference's producers emit
                                 // (CV_32F, utils.cpp:79-82) -- else NULL.  The plain fp64 kernels then fetch a stencil row
                                 // with ONE 16-byte load instead of two and widen it: same doubles, same results, half the
                                 // texture instructions and cache footprint.
  double dist[5];                // k1, k2, p1, p2, k3
  double A[9], d[3];             // second camera: b = A (R a' + t) + d,  [A d] = affine part of T12
  double Ai[9], di[3];           //                a' = Ai a + di,        [Ai di] = affine part of T12^-1
  float distf[5];
  float Af[9], df[3], Aif[9], dif[3];
};

// a group = the residual families (terms) that share one pose; its rows are contiguous
struct GroupDesc {
  int32_t tile_begin, tile_end;
  int32_t term_begin, term_end;
};

// Pose-dependent constants, rebuilt whenever a pose changes (by the host for ea_eval, by the
// device LM-step kernel inside ea_solve).  Uniform per problem -> scalar loads in the kernels.
struct PoseState {
  double q[4], t[3];
  double R[9];       // rotation actually applied (already transposed for the ROS flavour)
  double G[27];      // G[j] = d R / d delta_j, only read when unit_q == 0
  float Rf[9], tf[3], Gf[27];
  int32_t unit_q;    // | |q|^2 - 1 | <= 1e-12  -> J_delta = 2 (R a) x g
  int32_t active;    // 0: the problem's workgroups return immediately (solve finished)
  int32_t pad_[2];
};

static_assert(sizeof(ProblemDesc) % 8 == 0, "ProblemDesc is fetched as whole scalar dwords");

// Result of one reduced evaluation
struct EvalOut {
  double acc[kAccSlots];
};

}  // namespace ea

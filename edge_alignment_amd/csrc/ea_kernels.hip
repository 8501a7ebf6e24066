// ea_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the edge-alignment hot path.
//
// What the reference does per edge point on one CPU thread with dual numbers
// (ref: standalone/utils.h:48-80 EAResidue::operator(), ceres BiCubicInterpolator, AutoDiff,
// QuaternionParameterization, loss corrector) is done here by one lane per point:
//   b = R a + t  ->  (u,v) = K b / b_z  ->  16-tap Catmull-Rom sample of the DT image with
//   analytic d/du, d/dv  ->  analytic 1x6 row  ->  IRLS weight  ->  28 fp64 accumulators,
// followed by a wavefront-level transposing butterfly and a fixed-order cross-wave / cross-tile
// reduction (bit-reproducible: no float atomics).
//
// Data layout in HBM: points as SoA x[],y[],z[] (coalesced 4/8-byte loads per lane); the DT
// image row-major [v][u] with a 3-texel replicated border so that Grid2D's clamp-to-edge
// addressing needs no per-tap clamps; per-tile partial sums as 32 doubles (256 B) per tile.
// A workgroup handles a tile = run of consecutive points; their projected footprint (+halo) is
// staged through LDS when it fits, otherwise the taps come from L2 directly.
//
// No MFMA: pointwise arithmetic plus a 28-value reduction.

#include <hip/hip_runtime.h>

#include "ea_lm.h"
#include "ea_types.h"

namespace ea {

// ------------------------------------------------------------------------------------------------
// arithmetic helpers, T = float | double

template <typename T> __device__ __forceinline__ T t_rcp(T x);
template <> __device__ __forceinline__ float t_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> __device__ __forceinline__ double t_rcp<double>(double x) { return 1.0 / x; }

template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return __logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }

template <typename T> __device__ __forceinline__ T t_fma(T a, T b, T c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float t_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Catmull-Rom weights of the four taps at fraction x, and their derivatives
// (ceres CubicHermiteSpline written as tap weights).
template <typename T>
__device__ __forceinline__ void cr_weights(T x, T w[4], T d[4]) {
  const T x2 = x * x;
  w[0] = x * t_fma<T>(x, t_fma<T>(T(-0.5), x, T(1)), T(-0.5));
  w[1] = t_fma<T>(x2, t_fma<T>(T(1.5), x, T(-2.5)), T(1));
  w[2] = x * t_fma<T>(x, t_fma<T>(T(-1.5), x, T(2)), T(0.5));
  w[3] = x2 * t_fma<T>(T(0.5), x, T(-0.5));
  d[0] = t_fma<T>(x, t_fma<T>(T(-1.5), x, T(2)), T(-0.5));
  d[1] = x * t_fma<T>(T(4.5), x, T(-5));
  d[2] = t_fma<T>(x, t_fma<T>(T(-4.5), x, T(4)), T(0.5));
  d[3] = x * t_fma<T>(T(1.5), x, T(-1));
}

// per-point state carried from the projection phase to the sampling phase
template <typename T>
struct Proj {
  T bx, by, iz;   // warped point (x, y) and 1 / (b_z + z_eps)
  T cx_, cy_, cz_; // R a  (= b - t), for the unit-quaternion Jacobian
  T fu, fv;       // fractional parts of (u, v)
  int iu, iv;     // floor(u), floor(v), clamped to [-2, W] / [-2, H]
  int state;      // 0 = lane has no point, 1 = valid, 2 = functor returned false
};

template <typename T>
struct PoseT {
  T R[9], t[3];
};

template <typename T>
__device__ __forceinline__ void load_pose(const PoseState &ps, PoseT<T> &p) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p.R[i] = (T)ps.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) p.t[i] = (T)ps.t[i];
}

template <typename T>
__device__ __forceinline__ void project_point(const ProblemDesc &pd, const PoseT<T> &pose, T x, T y,
                                              T z, Proj<T> &o) {
  const T cxr = t_fma<T>(pose.R[2], z, t_fma<T>(pose.R[1], y, pose.R[0] * x));
  const T cyr = t_fma<T>(pose.R[5], z, t_fma<T>(pose.R[4], y, pose.R[3] * x));
  const T czr = t_fma<T>(pose.R[8], z, t_fma<T>(pose.R[7], y, pose.R[6] * x));
  const T bx = cxr + pose.t[0], by = cyr + pose.t[1], bz = czr + pose.t[2];
  const T zg = (T)pd.z_guard;
  // ref: utils.h:70-73 — `return false` inside (-0.01, 0.01)
  const bool bad = (zg > T(0)) && (bz < zg) && (bz > -zg);
  const T iz = t_rcp<T>(bz + (T)pd.z_eps);
  const T u = t_fma<T>((T)pd.fx * bx, iz, (T)pd.cx);
  const T v = t_fma<T>((T)pd.fy * by, iz, (T)pd.cy);
  // floor with saturation: beyond [-2, W] x [-2, H] every tap is the replicated border texel
  const T uf = floor(fmin(fmax(u, T(-2)), (T)pd.W));
  const T vf = floor(fmin(fmax(v, T(-2)), (T)pd.H));
  o.bx = bx; o.by = by; o.iz = iz;
  o.cx_ = cxr; o.cy_ = cyr; o.cz_ = czr;
  o.fu = u - uf;  // for saturated coordinates the taps are all equal and the fraction is irrelevant
  o.fv = v - vf;
  o.iu = (int)uf;
  o.iv = (int)vf;
  o.state = bad ? 2 : 1;
}

// 16 taps -> value and gradient.  taps(l, k) = DT(v = iv-1+l, u = iu-1+k)
template <typename T, typename TapFn>
__device__ __forceinline__ void bicubic(T fu, T fv, TapFn tap, T &f, T &Fu, T &Fv) {
  T wu[4], du[4], wv[4], dv[4];
  cr_weights<T>(fu, wu, du);
  cr_weights<T>(fv, wv, dv);
  f = T(0); Fu = T(0); Fv = T(0);
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const T p0 = tap(l, 0), p1 = tap(l, 1), p2 = tap(l, 2), p3 = tap(l, 3);
    const T rs = t_fma<T>(wu[3], p3, t_fma<T>(wu[2], p2, t_fma<T>(wu[1], p1, wu[0] * p0)));
    const T rd = t_fma<T>(du[3], p3, t_fma<T>(du[2], p2, t_fma<T>(du[1], p1, du[0] * p0)));
    f = t_fma<T>(wv[l], rs, f);
    Fv = t_fma<T>(dv[l], rs, Fv);
    Fu = t_fma<T>(wv[l], rd, Fu);
  }
}

// rho(s), rho'(s) — ceres loss_function.cc
template <typename T>
__device__ __forceinline__ void loss_eval(int kind, T a, T s, T &rho, T &w) {
  if (kind == 1) {  // Cauchy
    const T b = a * a;
    const T sum = t_fma<T>(s, t_rcp<T>(b), T(1));
    w = t_rcp<T>(sum);
    rho = b * t_log<T>(sum);
  } else if (kind == 2) {  // Huber
    const T b = a * a;
    if (s > b) {
      const T r = t_sqrt<T>(s);
      rho = t_fma<T>(T(2) * a, r, -b);
      w = a * t_rcp<T>(r);
    } else {
      rho = s; w = T(1);
    }
  } else {
    rho = s; w = T(1);
  }
}

// residual and raw 1x6 row of one point from its sample
template <typename T>
__device__ __forceinline__ void jacobian_row(const ProblemDesc &pd, const PoseState &ps,
                                             const Proj<T> &pr, T x, T y, T z, T Fu, T Fv, T J[6]) {
  const T gx = Fu * (T)pd.fx * pr.iz;
  const T gy = Fv * (T)pd.fy * pr.iz;
  const T gz = -t_fma<T>(gx, pr.bx, gy * pr.by) * pr.iz;
  if (ps.unit_q) {
    // d b / d delta = -2 [R a]x  =>  J_delta = 2 (R a) x g
    J[0] = T(2) * t_fma<T>(pr.cy_, gz, -(pr.cz_ * gy));
    J[1] = T(2) * t_fma<T>(pr.cz_, gx, -(pr.cx_ * gz));
    J[2] = T(2) * t_fma<T>(pr.cx_, gy, -(pr.cy_ * gx));
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double *G = ps.G + 9 * j;
      const T d0 = t_fma<T>((T)G[2], z, t_fma<T>((T)G[1], y, (T)G[0] * x));
      const T d1 = t_fma<T>((T)G[5], z, t_fma<T>((T)G[4], y, (T)G[3] * x));
      const T d2 = t_fma<T>((T)G[8], z, t_fma<T>((T)G[7], y, (T)G[6] * x));
      J[j] = t_fma<T>(gz, d2, t_fma<T>(gy, d1, gx * d0));
    }
  }
  J[3] = gx; J[4] = gy; J[5] = gz;
}

// ------------------------------------------------------------------------------------------------
// wavefront reduction of 32 values per lane: transposing butterfly.
// After the call lane L (0..63) holds in v[0] the wave total of slot
//   id(L) = 16*b0 + 8*b1 + 4*b2 + 2*b3 + b4   (b_k = bit k of L); lanes L and L^32 hold the same.
// 31 shuffles + 1 instead of 32*6.

__device__ __forceinline__ int butterfly_slot(int lane) {
  return ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
}

template <int M, int C>
__device__ __forceinline__ void butterfly_step(double (&v)[32], int lane) {
  const bool upper = (lane & M) != 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const double send = upper ? v[i] : v[i + C];
    const double keep = upper ? v[i + C] : v[i];
    v[i] = keep + __shfl_xor(send, M, 64);
  }
}

__device__ __forceinline__ void wave_reduce32(double (&v)[32], int lane) {
  butterfly_step<1, 16>(v, lane);
  butterfly_step<2, 8>(v, lane);
  butterfly_step<4, 4>(v, lane);
  butterfly_step<8, 2>(v, lane);
  butterfly_step<16, 1>(v, lane);
  v[0] += __shfl_xor(v[0], 32, 64);
}

__device__ __forceinline__ int wave_min_i32(int x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = min(x, __shfl_xor(x, m, 64));
  return x;
}
__device__ __forceinline__ int wave_max_i32(int x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = max(x, __shfl_xor(x, m, 64));
  return x;
}

// ------------------------------------------------------------------------------------------------
// fused evaluation kernel: residual + Jacobian + loss + JtJ/Jtr/cost partials per tile

constexpr int kRedBytes = 4 * kAccSlots * 8;  // cross-wave scratch: 4 waves x 32 doubles
constexpr int kHdrBytes = kRedBytes + 64;     // + bbox words, keeps the tile 16-byte aligned

template <typename T, int PPT>
__global__ __launch_bounds__(kBlockThreads) void ea_eval_fused_kernel(
    const ProblemDesc *__restrict__ probs, const Tile *__restrict__ tiles, int ntiles,
    int tiles_per_xcd, int xcd_remap, const PoseState *__restrict__ poses,
    double *__restrict__ partials, int lds_texels) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *s_red = reinterpret_cast<double *>(smem);
  int *s_box = reinterpret_cast<int *>(smem + kRedBytes);
  T *s_tile = reinterpret_cast<T *>(smem + kHdrBytes);

  // XCD-aware tile assignment: workgroups are dealt round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of tiles (neighbouring tiles read neighbouring image rows -> one L2).
  const int bid = blockIdx.x;
  const int tile_id = xcd_remap ? (bid & 7) * tiles_per_xcd + (bid >> 3) : bid;
  if (tile_id >= ntiles) return;
  const Tile tile = tiles[tile_id];
  const ProblemDesc &pd = probs[tile.problem];
  const PoseState &ps = poses[tile.problem];
  if (!ps.active) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  PoseT<T> pose;
  load_pose<T>(ps, pose);
  const T *__restrict__ px = static_cast<const T *>(pd.x);
  const T *__restrict__ py = static_cast<const T *>(pd.y);
  const T *__restrict__ pz = static_cast<const T *>(pd.z);

  // ---- phase 1: coalesced point loads, warp + projection
  Proj<T> pr[PPT];
  T X[PPT], Y[PPT], Z[PPT];
  int bb_u0 = 0x7fffffff, bb_u1 = -0x7fffffff, bb_v0 = 0x7fffffff, bb_v1 = -0x7fffffff;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int j = tid + k * kBlockThreads;
    pr[k].state = 0;
    if (j < tile.count) {
      const int i = tile.start + j;
      X[k] = px[i]; Y[k] = py[i]; Z[k] = pz[i];
      project_point<T>(pd, pose, X[k], Y[k], Z[k], pr[k]);
      if (pr[k].state == 1) {
        bb_u0 = min(bb_u0, pr[k].iu); bb_u1 = max(bb_u1, pr[k].iu);
        bb_v0 = min(bb_v0, pr[k].iv); bb_v1 = max(bb_v1, pr[k].iv);
      }
    }
  }

  // ---- phase 2: footprint of the tile, staged through LDS when it fits
  bool use_lds = false;
  int u0 = 0, v0 = 0, tw = 0;
  if (lds_texels > 0) {
    bb_u0 = wave_min_i32(bb_u0); bb_u1 = wave_max_i32(bb_u1);
    bb_v0 = wave_min_i32(bb_v0); bb_v1 = wave_max_i32(bb_v1);
    if (lane == 0) {
      s_box[4 * wave + 0] = bb_u0; s_box[4 * wave + 1] = bb_u1;
      s_box[4 * wave + 2] = bb_v0; s_box[4 * wave + 3] = bb_v1;
    }
    __syncthreads();
    int U0 = s_box[0], U1 = s_box[1], V0 = s_box[2], V1 = s_box[3];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      U0 = min(U0, s_box[4 * w + 0]); U1 = max(U1, s_box[4 * w + 1]);
      V0 = min(V0, s_box[4 * w + 2]); V1 = max(V1, s_box[4 * w + 3]);
    }
    if (U1 >= U0) {
      u0 = U0 - 1; v0 = V0 - 1;
      tw = U1 - U0 + 4;
      const int th = V1 - V0 + 4;
      const long long area = (long long)tw * (long long)th;
      if (area <= (long long)lds_texels) {
        use_lds = true;
        const T *__restrict__ img = static_cast<const T *>(pd.dt) +
                                    (size_t)(v0 + kImagePad) * (size_t)pd.pitch + (u0 + kImagePad);
        const float inv_tw = 1.0f / (float)tw;
        for (int idx = tid; idx < (int)area; idx += kBlockThreads) {
          int row = (int)((float)idx * inv_tw);
          int col = idx - row * tw;
          if (col < 0) { row -= 1; col += tw; }
          if (col >= tw) { row += 1; col -= tw; }
          s_tile[idx] = img[(size_t)row * (size_t)pd.pitch + col];
        }
        __syncthreads();
      }
    }
  }

  // ---- phase 3: sample, Jacobian, weights, accumulate
  T acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = T(0);
  int n_bad = 0;
  const T *__restrict__ gimg = static_cast<const T *>(pd.dt) + (size_t)kImagePad * (size_t)pd.pitch + kImagePad;
  const int pitch = pd.pitch;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (pr[k].state == 2) n_bad += 1;
    if (pr[k].state != 1) continue;
    T f, Fu, Fv;
    if (use_lds) {
      const T *base = s_tile + (pr[k].iv - 1 - v0) * tw + (pr[k].iu - 1 - u0);
      const int stride = tw;
      bicubic<T>(pr[k].fu, pr[k].fv, [&](int l, int c) { return base[l * stride + c]; }, f, Fu, Fv);
    } else {
      const T *base = gimg + (ptrdiff_t)(pr[k].iv - 1) * pitch + (pr[k].iu - 1);
      bicubic<T>(pr[k].fu, pr[k].fv, [&](int l, int c) { return base[(ptrdiff_t)l * pitch + c]; }, f, Fu, Fv);
    }
    T J[6];
    jacobian_row<T>(pd, ps, pr[k], X[k], Y[k], Z[k], Fu, Fv, J);
    T rho, w;
    loss_eval<T>(pd.loss_kind, (T)pd.loss_a, f * f, rho, w);
    const T wr = w * f;
    int s = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const T wJa = w * J[a];
#pragma unroll
      for (int b = a; b < 6; ++b) { acc[s] = t_fma<T>(wJa, J[b], acc[s]); ++s; }
      acc[kAccJtr + a] = t_fma<T>(J[a], wr, acc[kAccJtr + a]);
    }
    acc[kAccCost] = t_fma<T>(T(0.5), rho, acc[kAccCost]);
  }

  // ---- phase 4: wavefront butterfly, then fixed-order cross-wave sum
  double v[32];
#pragma unroll
  for (int i = 0; i < 28; ++i) v[i] = (double)acc[i];
  v[28] = (double)n_bad; v[29] = 0.0; v[30] = 0.0; v[31] = 0.0;
  wave_reduce32(v, lane);
  __syncthreads();  // s_tile / s_box readers are done; s_red is a separate region but keep phases ordered
  if (lane < 32) s_red[wave * kAccSlots + butterfly_slot(lane)] = v[0];
  __syncthreads();
  if (tid < kAccSlots) {
    const double sum = ((s_red[tid] + s_red[kAccSlots + tid]) + s_red[2 * kAccSlots + tid]) + s_red[3 * kAccSlots + tid];
    partials[(size_t)tile_id * kAccSlots + tid] = sum;
  }
}

// ------------------------------------------------------------------------------------------------
// per-point outputs (parity / "EAResidue batch Evaluate" view): r[n], J[n*6]

template <typename T>
__global__ __launch_bounds__(kBlockThreads) void ea_eval_points_kernel(
    const ProblemDesc *__restrict__ probs, int problem, const PoseState *__restrict__ poses,
    double *__restrict__ r_out, double *__restrict__ J_out, int corrected) {
  const ProblemDesc &pd = probs[problem];
  const PoseState &ps = poses[problem];
  const int i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= pd.n) return;
  PoseT<T> pose;
  load_pose<T>(ps, pose);
  const T x = static_cast<const T *>(pd.x)[i], y = static_cast<const T *>(pd.y)[i], z = static_cast<const T *>(pd.z)[i];
  Proj<T> pr;
  project_point<T>(pd, pose, x, y, z, pr);
  const double nan = __builtin_nan("");
  if (pr.state != 1) {
    if (r_out) r_out[i] = nan;
    if (J_out)
      for (int a = 0; a < 6; ++a) J_out[(size_t)i * 6 + a] = nan;
    return;
  }
  const T *base = static_cast<const T *>(pd.dt) + (size_t)kImagePad * (size_t)pd.pitch + kImagePad +
                  (ptrdiff_t)(pr.iv - 1) * pd.pitch + (pr.iu - 1);
  const int pitch = pd.pitch;
  T f, Fu, Fv;
  bicubic<T>(pr.fu, pr.fv, [&](int l, int c) { return base[(ptrdiff_t)l * pitch + c]; }, f, Fu, Fv);
  T J[6];
  jacobian_row<T>(pd, ps, pr, x, y, z, Fu, Fv, J);
  T sc = T(1);
  if (corrected) {
    T rho, w;
    loss_eval<T>(pd.loss_kind, (T)pd.loss_a, f * f, rho, w);
    sc = t_sqrt<T>(w);
  }
  if (r_out) r_out[i] = (double)(sc * f);
  if (J_out)
    for (int a = 0; a < 6; ++a) J_out[(size_t)i * 6 + a] = (double)(sc * J[a]);
}

// ------------------------------------------------------------------------------------------------
// fixed-order reduction of a problem's tile partials -> 32 accumulators

__device__ __forceinline__ void reduce_tiles(const double *__restrict__ partials, int tile_begin,
                                             int tile_end, double *s_part /* 8 x 32 */,
                                             double *out /* 32, lanes 0..31 write */) {
  const int tid = threadIdx.x;
  const int id = tid & 31, j = tid >> 5;  // 8 strided partial sums per slot
  double s = 0.0;
  for (int tI = tile_begin + j; tI < tile_end; tI += 8) s += partials[(size_t)tI * kAccSlots + id];
  s_part[j * kAccSlots + id] = s;
  __syncthreads();
  if (tid < kAccSlots) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += s_part[k * kAccSlots + tid];
    out[tid] = tot;
  }
}

__global__ __launch_bounds__(kBlockThreads) void ea_reduce_kernel(const ProblemDesc *__restrict__ probs,
                                                                  const double *__restrict__ partials,
                                                                  EvalOut *__restrict__ out) {
  __shared__ double s_part[8 * kAccSlots];
  const ProblemDesc &pd = probs[blockIdx.x];
  reduce_tiles(partials, pd.tile_begin, pd.tile_end, s_part, out[blockIdx.x].acc);
}

// LM step: reduce this problem's partials, advance the trust-region state machine, publish the
// next pose to evaluate.  One workgroup per problem; the state machine itself is scalar work on
// lane 0 (6x6 algebra in fp64).
__global__ __launch_bounds__(kBlockThreads) void ea_lm_step_kernel(
    const ProblemDesc *__restrict__ probs, const double *__restrict__ partials,
    PoseState *__restrict__ poses, LMState *__restrict__ states, LMOptions opt,
    int *__restrict__ running_flags) {
  __shared__ double s_part[8 * kAccSlots];
  __shared__ double s_acc[kAccSlots];
  const int p = blockIdx.x;
  LMState *st = states + p;
  if (!st->running) return;  // uniform
  const ProblemDesc &pd = probs[p];
  reduce_tiles(partials, pd.tile_begin, pd.tile_end, s_part, s_acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc[kAccSlots];
    for (int i = 0; i < kAccSlots; ++i) acc[i] = s_acc[i];
    if (st->num_evals == 0) lm_begin(st, &opt, acc);
    else lm_advance(st, &opt, acc);
    make_pose_state(st->cand, st->rot_transposed, st->running, poses + p);
    running_flags[p] = st->running;
  }
}

// pad + convert a row-major [H][W] device image into the replicated-border layout
template <typename T>
__global__ void ea_pad_image_kernel(const T *__restrict__ src, int H, int W, T *__restrict__ dst, int pitch) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;  // padded coords
  const int v = blockIdx.y;
  if (u >= W + 2 * kImagePad) return;
  const int su = min(max(u - kImagePad, 0), W - 1);
  const int sv = min(max(v - kImagePad, 0), H - 1);
  dst[(size_t)v * pitch + u] = src[(size_t)sv * W + su];
}

// ------------------------------------------------------------------------------------------------
// launchers (called from ea_capi.cpp)

hipError_t launch_eval_fused(int dtype, int ppt, const ProblemDesc *probs, const Tile *tiles, int ntiles,
                             int xcd_remap, const PoseState *poses, double *partials, int lds_bytes,
                             hipStream_t stream) {
  if (ntiles <= 0) return hipSuccess;
  const int tiles_per_xcd = (ntiles + 7) / 8;
  const int grid = xcd_remap ? tiles_per_xcd * 8 : ntiles;
  const int esz = dtype == 1 ? 4 : 8;
  const int lds_texels = lds_bytes > 0 ? lds_bytes / esz : 0;
  const size_t shmem = (size_t)kHdrBytes + (size_t)lds_texels * esz;
#define EA_LAUNCH(T, P)                                                                         \
  hipLaunchKernelGGL((ea_eval_fused_kernel<T, P>), dim3(grid), dim3(kBlockThreads), shmem, stream, probs, \
                     tiles, ntiles, tiles_per_xcd, xcd_remap, poses, partials, lds_texels)
  if (dtype == 1) {
    if (ppt == 1) EA_LAUNCH(float, 1);
    else if (ppt == 2) EA_LAUNCH(float, 2);
    else EA_LAUNCH(float, 4);
  } else {
    if (ppt == 1) EA_LAUNCH(double, 1);
    else if (ppt == 2) EA_LAUNCH(double, 2);
    else EA_LAUNCH(double, 4);
  }
#undef EA_LAUNCH
  return hipGetLastError();
}

hipError_t launch_eval_points(int dtype, const ProblemDesc *probs, int problem, int n, const PoseState *poses,
                              double *r_out, double *J_out, int corrected, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int grid = (n + kBlockThreads - 1) / kBlockThreads;
  if (dtype == 1)
    hipLaunchKernelGGL((ea_eval_points_kernel<float>), dim3(grid), dim3(kBlockThreads), 0, stream, probs, problem,
                       poses, r_out, J_out, corrected);
  else
    hipLaunchKernelGGL((ea_eval_points_kernel<double>), dim3(grid), dim3(kBlockThreads), 0, stream, probs, problem,
                       poses, r_out, J_out, corrected);
  return hipGetLastError();
}

hipError_t launch_reduce(const ProblemDesc *probs, int count, const double *partials, EvalOut *out,
                         hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(ea_reduce_kernel, dim3(count), dim3(kBlockThreads), 0, stream, probs, partials, out);
  return hipGetLastError();
}

hipError_t launch_lm_step(const ProblemDesc *probs, int count, const double *partials, PoseState *poses,
                          LMState *states, const LMOptions &opt, int *running_flags, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(ea_lm_step_kernel, dim3(count), dim3(kBlockThreads), 0, stream, probs, partials, poses,
                     states, opt, running_flags);
  return hipGetLastError();
}

hipError_t launch_pad_image(int dtype, const void *src, int H, int W, void *dst, int pitch, hipStream_t stream) {
  dim3 block(256), grid((W + 2 * kImagePad + 255) / 256, H + 2 * kImagePad);
  if (dtype == 1)
    hipLaunchKernelGGL((ea_pad_image_kernel<float>), grid, block, 0, stream, (const float *)src, H, W, (float *)dst, pitch);
  else
    hipLaunchKernelGGL((ea_pad_image_kernel<double>), grid, block, 0, stream, (const double *)src, H, W, (double *)dst, pitch);
  return hipGetLastError();
}

}  // namespace ea

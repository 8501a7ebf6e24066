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
// staged through LDS when it fits, otherwise the taps come from L2 directly (four unaligned
// 16-byte row loads per point).
//
// No MFMA: pointwise arithmetic plus a 28-value reduction.

#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#if defined(EA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
namespace ea {
extern __device__ unsigned long long *g_lm_stamp_buf;
extern __device__ int g_lm_probe_row;
}
#define EA_LM_PROBE(k)                                                                              \
  do {                                                                                              \
    if (g_lm_stamp_buf && blockIdx.x == 0) {                                                        \
      unsigned long long t_;                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      g_lm_stamp_buf[64 * 8 + g_lm_probe_row * 8 + (k)] = t_;                                       \
    }                                                                                               \
  } while (0)
#endif
#include "ea_lm.h"
#include "ea_types.h"

namespace ea {

// Diagnostic build only (-DEA_STAMPS, scripts/build_stamps.sh -> lib/libea_hip_stamps.so): wave 0 of every
// workgroup stamps s_memtime at the phase boundaries into a buffer of its own; no output value depends on it.
#ifdef EA_STAMPS
__device__ unsigned long long *g_stamp_buf = nullptr;
#ifndef EA_STAMPS_EVAL
// (the evaluation kernel's stamps need -DEA_STAMPS_EVAL on top: since the kernel head keeps its uniforms in named SGPRs the
// stamped form of it no longer compiles on this toolchain -- "illegal VGPR to SGPR copy"; the LM-step stamps are unaffected)
#define EA_STAMP(slot) do {} while (0)
#else
#define EA_STAMP(slot)                                                                              \
  do {                                                                                              \
    if (g_stamp_buf && threadIdx.x == 0) {                                                          \
      unsigned long long t_;                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      g_stamp_buf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = t_;                 \
    }                                                                                               \
  } while (0)
#endif
__device__ unsigned long long *g_lm_stamp_buf = nullptr;  // [64 evaluations][8] kernel + [64][8] state-machine probes, problem 0
__device__ int g_lm_probe_row = 0;
#define EA_LM_STAMP(slot, eval)                                                                     \
  do {                                                                                              \
    if (g_lm_stamp_buf && threadIdx.x == 0 && blockIdx.x == 0) {                                    \
      unsigned long long t_;                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      g_lm_stamp_buf[((eval) & 63) * 8 + (slot)] = t_;                                              \
    }                                                                                               \
  } while (0)
// (a stamp taken before its row index is known: read the clock now, store later)
#define EA_LM_CLOCK(var)                                                                            \
  unsigned long long var = 0;                                                                       \
  do {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#define EA_LM_STAMP_PUT(slot, eval, var)                                                            \
  do {                                                                                              \
    if (g_lm_stamp_buf && threadIdx.x == 0 && blockIdx.x == 0) g_lm_stamp_buf[((eval) & 63) * 8 + (slot)] = var; \
  } while (0)
#else
#define EA_STAMP(slot) do {} while (0)
#define EA_LM_STAMP(slot, eval) do {} while (0)
#define EA_LM_CLOCK(var) do {} while (0)
#define EA_LM_STAMP_PUT(slot, eval, var) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// arithmetic helpers, T = float | double

template <typename T> __device__ __forceinline__ T t_rcp(T x);
template <> __device__ __forceinline__ float t_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
// v_rcp_f64 + two Newton steps: 5 instructions and <= 1 ulp instead of the ~14 of an IEEE division (three per point)
template <> __device__ __forceinline__ double t_rcp<double>(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

template <typename T> __device__ __forceinline__ T t_log(T x);
// v_log_f32 is log2 and needs no denormal pre-scaling here: its only caller passes 1 + s / a^2 >= 1
template <> __device__ __forceinline__ float t_log<float>(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
// fp64 log for the Cauchy loss, argument 1 + s / a^2 in [1, inf): frexp to m in [sqrt(1/2), sqrt(2)), then
// log m = 2 atanh z, z = (m - 1) / (m + 1), |z| <= 0.1716, as the odd series to z^21 (next term < 1e-18 relative).
// ~33 instructions and ~2 ulp, against ~110 for libm's log (which was a fifth of the fp64 kernel's vector work).
template <> __device__ __forceinline__ double t_log<double>(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = __builtin_amdgcn_ldexp(m, lo ? 1 : 0);
  e -= lo ? 1 : 0;
  const double z = (m - 1.0) * t_rcp<double>(m + 1.0);
  const double z2 = z * z;
  double p = 2.0 / 21.0;
  p = __builtin_fma(p, z2, 2.0 / 19.0);
  p = __builtin_fma(p, z2, 2.0 / 17.0);
  p = __builtin_fma(p, z2, 2.0 / 15.0);
  p = __builtin_fma(p, z2, 2.0 / 13.0);
  p = __builtin_fma(p, z2, 2.0 / 11.0);
  p = __builtin_fma(p, z2, 2.0 / 9.0);
  p = __builtin_fma(p, z2, 2.0 / 7.0);
  p = __builtin_fma(p, z2, 2.0 / 5.0);
  p = __builtin_fma(p, z2, 2.0 / 3.0);
  const double lm = __builtin_fma(z * z2, p, z + z);
  const double ef = (double)e;
  // ln 2 split so that e * hi is exact for |e| < 2^11
  return __builtin_fma(ef, 0.693147180369123816490, __builtin_fma(ef, 1.90821492927058770002e-10, lm));
}

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

// uniform (per-problem / per-pose) values in the kernel's arithmetic type
template <typename T> struct Uni;
template <> struct Uni<double> {
  static __device__ __forceinline__ const double *R(const PoseState &ps) { return ps.R; }
  static __device__ __forceinline__ const double *t(const PoseState &ps) { return ps.t; }
  static __device__ __forceinline__ const double *G(const PoseState &ps) { return ps.G; }
  static __device__ __forceinline__ double fx(const ProblemDesc &pd) { return pd.fx; }
  static __device__ __forceinline__ double fy(const ProblemDesc &pd) { return pd.fy; }
  static __device__ __forceinline__ double cx(const ProblemDesc &pd) { return pd.cx; }
  static __device__ __forceinline__ double cy(const ProblemDesc &pd) { return pd.cy; }
  static __device__ __forceinline__ double loss_a(const ProblemDesc &pd) { return pd.loss_a; }
  static __device__ __forceinline__ double loss_inv_b(const ProblemDesc &pd) { return pd.loss_inv_b; }
  static __device__ __forceinline__ double z_guard(const ProblemDesc &pd) { return pd.z_guard; }
  static __device__ __forceinline__ double z_eps(const ProblemDesc &pd) { return pd.z_eps; }
  static __device__ __forceinline__ const double *dist(const ProblemDesc &pd) { return pd.dist; }
  static __device__ __forceinline__ const double *A(const ProblemDesc &pd) { return pd.A; }
  static __device__ __forceinline__ const double *d(const ProblemDesc &pd) { return pd.d; }
  static __device__ __forceinline__ const double *Ai(const ProblemDesc &pd) { return pd.Ai; }
  static __device__ __forceinline__ const double *di(const ProblemDesc &pd) { return pd.di; }
};
template <> struct Uni<float> {
  static __device__ __forceinline__ const float *R(const PoseState &ps) { return ps.Rf; }
  static __device__ __forceinline__ const float *t(const PoseState &ps) { return ps.tf; }
  static __device__ __forceinline__ const float *G(const PoseState &ps) { return ps.Gf; }
  static __device__ __forceinline__ float fx(const ProblemDesc &pd) { return pd.fxf; }
  static __device__ __forceinline__ float fy(const ProblemDesc &pd) { return pd.fyf; }
  static __device__ __forceinline__ float cx(const ProblemDesc &pd) { return pd.cxf; }
  static __device__ __forceinline__ float cy(const ProblemDesc &pd) { return pd.cyf; }
  static __device__ __forceinline__ float loss_a(const ProblemDesc &pd) { return pd.loss_af; }
  static __device__ __forceinline__ float loss_inv_b(const ProblemDesc &pd) { return pd.loss_inv_bf; }
  static __device__ __forceinline__ float z_guard(const ProblemDesc &pd) { return pd.z_guardf; }
  static __device__ __forceinline__ float z_eps(const ProblemDesc &pd) { return pd.z_epsf; }
  static __device__ __forceinline__ const float *dist(const ProblemDesc &pd) { return pd.distf; }
  static __device__ __forceinline__ const float *A(const ProblemDesc &pd) { return pd.Af; }
  static __device__ __forceinline__ const float *d(const ProblemDesc &pd) { return pd.df; }
  static __device__ __forceinline__ const float *Ai(const ProblemDesc &pd) { return pd.Aif; }
  static __device__ __forceinline__ const float *di(const ProblemDesc &pd) { return pd.dif; }
};

// The pose as the fused kernel carries it: rotation and translation in the kernel's arithmetic type, held in SGPRs
// (fetched speculatively in the same batch as the descriptor), the rarely needed rest through a pointer.
template <typename T>
struct PoseLite {
  T R[9], t[3];
  int unit_q;
  const PoseState *full;
};
template <typename T> struct PoseAcc {
  static __device__ __forceinline__ const T *R(const PoseState &ps) { return Uni<T>::R(ps); }
  static __device__ __forceinline__ const T *t(const PoseState &ps) { return Uni<T>::t(ps); }
  static __device__ __forceinline__ const T *G(const PoseState &ps) { return Uni<T>::G(ps); }
  static __device__ __forceinline__ int unit_q(const PoseState &ps) { return ps.unit_q; }
  static __device__ __forceinline__ const T *R(const PoseLite<T> &ps) { return ps.R; }
  static __device__ __forceinline__ const T *t(const PoseLite<T> &ps) { return ps.t; }
  static __device__ __forceinline__ const T *G(const PoseLite<T> &ps) { return Uni<T>::G(*ps.full); }
  static __device__ __forceinline__ int unit_q(const PoseLite<T> &ps) { return ps.unit_q; }
};

// per-point state carried from the projection phase to the sampling phase
template <typename T>
struct Proj {
  T bx, by, iz;    // warped point (x, y) and 1 / (b_z + z_eps)
  T fxz, fyz;      // fx * iz, fy * iz
  T cx_, cy_, cz_; // R a  (= b - t), for the unit-quaternion Jacobian
  T fu, fv;        // fractional parts of (u, v)
  int iu, iv;      // floor(u), floor(v), saturated to [-2, W] / [-2, H]
  int state;       // 0 = lane has no point, 1 = valid, 2 = functor returned false
};

template <typename T, typename PS>
__device__ __forceinline__ void project_point(const ProblemDesc &pd, const PS &ps, T x, T y, T z,
                                              Proj<T> &o) {
  const T *R = PoseAcc<T>::R(ps);
  const T *t = PoseAcc<T>::t(ps);
  const T cxr = t_fma<T>(R[2], z, t_fma<T>(R[1], y, R[0] * x));
  const T cyr = t_fma<T>(R[5], z, t_fma<T>(R[4], y, R[3] * x));
  const T czr = t_fma<T>(R[8], z, t_fma<T>(R[7], y, R[6] * x));
  const T bx = cxr + t[0], by = cyr + t[1], bz = czr + t[2];
  const T zg = Uni<T>::z_guard(pd);
  // ref: utils.h:70-73 — `return false` inside (-0.01, 0.01)
  const bool bad = (zg > T(0)) && (bz < zg) && (bz > -zg);
  const T iz = t_rcp<T>(bz + Uni<T>::z_eps(pd));
  // fx / b_z and fy / b_z serve the projection here and the gradient in jacobian_row
  const T fxz = Uni<T>::fx(pd) * iz, fyz = Uni<T>::fy(pd) * iz;
  const T u = t_fma<T>(fxz, bx, Uni<T>::cx(pd));
  const T v = t_fma<T>(fyz, by, Uni<T>::cy(pd));
  // Texel index saturates: beyond [-2, W] x [-2, H] every tap is the replicated border texel.
  // The fraction stays the true one (Ceres: r - int(floor(r))): with equal taps the spline is
  // constant only while the weights stay O(1).
  const T uf = floor(u), vf = floor(v);
  o.bx = bx; o.by = by; o.iz = iz;
  o.fxz = fxz; o.fyz = fyz;
  o.cx_ = cxr; o.cy_ = cyr; o.cz_ = czr;
  o.fu = u - uf;
  o.fv = v - vf;
  o.iu = (int)fmin(fmax(uf, T(-2)), (T)pd.W);
  o.iv = (int)fmin(fmax(vf, T(-2)), (T)pd.H);
  o.state = bad ? 2 : 1;
}

// ---- residual variants: EAResidueEx (Brown-Conrady distortion, utils.h:102-177),
// EAResidueSecondCam (second camera of a rigid rig, b_T_a' = T12 b_T_a T12^-1, utils.h:179-292) and both
// (utils.h:295-421).  Same skeleton as the plain functor with an affine map either side of the pose and the
// distortion chain rule in the gradient.
template <typename T>
struct ProjV {
  T x, y, iz;       // normalised image coordinates and 1 / (b_z + z_eps)
  T ax, ay, az;     // a' = T12^-1 a (the point the pose acts on)
  T cx_, cy_, cz_;  // R a'
  T xdx, xdy, ydx, ydy;  // d(distorted x,y) / d(x,y)
  T fu, fv;
  int iu, iv;
  int state;
};

template <typename T, typename PS>
__device__ __forceinline__ void project_point_var(const ProblemDesc &pd, const PS &ps, T px, T py, T pz,
                                                  ProjV<T> &o) {
  const T *R = PoseAcc<T>::R(ps);
  const T *t = PoseAcc<T>::t(ps);
  T ax = px, ay = py, az = pz;
  if (pd.variant & 2) {
    const T *Ai = Uni<T>::Ai(pd), *di = Uni<T>::di(pd);
    ax = t_fma<T>(Ai[2], pz, t_fma<T>(Ai[1], py, Ai[0] * px)) + di[0];
    ay = t_fma<T>(Ai[5], pz, t_fma<T>(Ai[4], py, Ai[3] * px)) + di[1];
    az = t_fma<T>(Ai[8], pz, t_fma<T>(Ai[7], py, Ai[6] * px)) + di[2];
  }
  const T cxr = t_fma<T>(R[2], az, t_fma<T>(R[1], ay, R[0] * ax));
  const T cyr = t_fma<T>(R[5], az, t_fma<T>(R[4], ay, R[3] * ax));
  const T czr = t_fma<T>(R[8], az, t_fma<T>(R[7], ay, R[6] * ax));
  T bx = cxr + t[0], by = cyr + t[1], bz = czr + t[2];
  if (pd.variant & 2) {
    const T *A = Uni<T>::A(pd), *d = Uni<T>::d(pd);
    const T c0 = bx, c1 = by, c2 = bz;
    bx = t_fma<T>(A[2], c2, t_fma<T>(A[1], c1, A[0] * c0)) + d[0];
    by = t_fma<T>(A[5], c2, t_fma<T>(A[4], c1, A[3] * c0)) + d[1];
    bz = t_fma<T>(A[8], c2, t_fma<T>(A[7], c1, A[6] * c0)) + d[2];
  }
  const T zg = Uni<T>::z_guard(pd);
  const bool bad = (zg > T(0)) && (bz < zg) && (bz > -zg);
  const T iz = t_rcp<T>(bz + Uni<T>::z_eps(pd));
  const T x = bx * iz, y = by * iz;
  T xd = x, yd = y;
  o.xdx = T(1); o.xdy = T(0); o.ydx = T(0); o.ydy = T(1);
  if (pd.variant & 1) {
    const T *k = Uni<T>::dist(pd);  // k1, k2, p1, p2, k3
    const T r2 = t_fma<T>(x, x, y * y), r4 = r2 * r2, r6 = r4 * r2;
    const T D = t_fma<T>(k[4], r6, t_fma<T>(k[1], r4, t_fma<T>(k[0], r2, T(1))));
    const T Dp = t_fma<T>(T(3) * k[4], r4, t_fma<T>(T(2) * k[1], r2, k[0]));
    const T xy = x * y;
    xd = t_fma<T>(k[3], t_fma<T>(T(2) * x, x, r2), t_fma<T>(T(2) * k[2], xy, x * D));
    yd = t_fma<T>(k[2], t_fma<T>(T(2) * y, y, r2), t_fma<T>(T(2) * k[3], xy, y * D));
    o.xdx = t_fma<T>(T(6) * k[3], x, t_fma<T>(T(2) * k[2], y, t_fma<T>(T(2) * x * x, Dp, D)));
    o.xdy = t_fma<T>(T(2) * k[3], y, t_fma<T>(T(2) * k[2], x, T(2) * xy * Dp));
    o.ydx = t_fma<T>(T(2) * k[2], x, t_fma<T>(T(2) * k[3], y, T(2) * xy * Dp));
    o.ydy = t_fma<T>(T(6) * k[2], y, t_fma<T>(T(2) * k[3], x, t_fma<T>(T(2) * y * y, Dp, D)));
  }
  const T u = t_fma<T>(Uni<T>::fx(pd), xd, Uni<T>::cx(pd));
  const T v = t_fma<T>(Uni<T>::fy(pd), yd, Uni<T>::cy(pd));
  const T uf = floor(u), vf = floor(v);
  o.x = x; o.y = y; o.iz = iz;
  o.ax = ax; o.ay = ay; o.az = az;
  o.cx_ = cxr; o.cy_ = cyr; o.cz_ = czr;
  // Where the four taps along an axis are the one replicated border texel (floor <= -2 or >= extent), Ceres' Horner
  // spline returns that texel and a derivative of exactly 0.  The tap-weight form used here returns texel * (sum of the
  // derivative weights) = texel * O(eps) instead, which the distortion chain rule then multiplies by d(u,v)/d(x,y):
  // unbounded for points far outside the field of view (r^6 terms; 1e6 .. 1e12 seen in scripts/soak_variants.py, i.e.
  // rows of O(1..10) where the reference has zeros).  A zero fraction makes the weights (0,1,0,0) / (-.5,0,.5,0) and
  // both results exact.  (The plain functor's chain factor is fx / b_z <= fx / z_guard, so it keeps the true fraction.)
  o.fu = (uf <= T(-2) || uf >= (T)pd.W) ? T(0) : u - uf;
  o.fv = (vf <= T(-2) || vf >= (T)pd.H) ? T(0) : v - vf;
  o.iu = (int)fmin(fmax(uf, T(-2)), (T)pd.W);
  o.iv = (int)fmin(fmax(vf, T(-2)), (T)pd.H);
  o.state = bad ? 2 : 1;
}

template <typename T, typename PS>
__device__ __forceinline__ void jacobian_row_var(const ProblemDesc &pd, const PS &ps, const ProjV<T> &pr,
                                                 T Fu, T Fv, T J[6]) {
  const T fu_ = Fu * Uni<T>::fx(pd), fv_ = Fv * Uni<T>::fy(pd);
  const T rx = t_fma<T>(fv_, pr.ydx, fu_ * pr.xdx);  // d r / d x
  const T ry = t_fma<T>(fv_, pr.ydy, fu_ * pr.xdy);  // d r / d y
  T gx = rx * pr.iz, gy = ry * pr.iz, gz = -t_fma<T>(rx, pr.x, ry * pr.y) * pr.iz;  // d r / d b
  if (pd.variant & 2) {  // back through b = A c + d:  g_c = A^T g_b
    const T *A = Uni<T>::A(pd);
    const T g0 = t_fma<T>(A[6], gz, t_fma<T>(A[3], gy, A[0] * gx));
    const T g1 = t_fma<T>(A[7], gz, t_fma<T>(A[4], gy, A[1] * gx));
    const T g2 = t_fma<T>(A[8], gz, t_fma<T>(A[5], gy, A[2] * gx));
    gx = g0; gy = g1; gz = g2;
  }
  if (PoseAcc<T>::unit_q(ps)) {
    J[0] = T(2) * t_fma<T>(pr.cy_, gz, -(pr.cz_ * gy));
    J[1] = T(2) * t_fma<T>(pr.cz_, gx, -(pr.cx_ * gz));
    J[2] = T(2) * t_fma<T>(pr.cx_, gy, -(pr.cy_ * gx));
  } else {
    // (the empty asm keeps this uniform branch a branch: left alone, the compiler computes the 30 extra FMAs of the
    // general path for every point and selects afterwards)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T *G = PoseAcc<T>::G(ps) + 9 * j;
      const T d0 = t_fma<T>(G[2], pr.az, t_fma<T>(G[1], pr.ay, G[0] * pr.ax));
      const T d1 = t_fma<T>(G[5], pr.az, t_fma<T>(G[4], pr.ay, G[3] * pr.ax));
      const T d2 = t_fma<T>(G[8], pr.az, t_fma<T>(G[7], pr.ay, G[6] * pr.ax));
      J[j] = t_fma<T>(gz, d2, t_fma<T>(gy, d1, gx * d0));
    }
  }
  J[3] = gx; J[4] = gy; J[5] = gz;
}

// four consecutive texels of one image row; 4/8-byte aligned only (the hardware's unaligned
// access mode turns this into one or two dwordx4 loads instead of four scalar ones)
template <typename T>
struct __attribute__((packed, aligned(sizeof(T)))) Row4 {
  T p0, p1, p2, p3;
};

// 16 taps -> value and gradient.  row(l) = DT(v = iv-1+l, u = iu-1 .. iu+2)
template <typename T, typename RowFn>
__device__ __forceinline__ void bicubic(T fu, T fv, RowFn row, T &f, T &Fu, T &Fv) {
  T wu[4], du[4], wv[4], dv[4];
  cr_weights<T>(fu, wu, du);
  cr_weights<T>(fv, wv, dv);
  f = T(0); Fu = T(0); Fv = T(0);
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const Row4<T> r = row(l);
    const T rs = t_fma<T>(wu[3], r.p3, t_fma<T>(wu[2], r.p2, t_fma<T>(wu[1], r.p1, wu[0] * r.p0)));
    const T rd = t_fma<T>(du[3], r.p3, t_fma<T>(du[2], r.p2, t_fma<T>(du[1], r.p1, du[0] * r.p0)));
    f = t_fma<T>(wv[l], rs, f);
    Fv = t_fma<T>(dv[l], rs, Fv);
    Fu = t_fma<T>(wv[l], rd, Fu);
  }
}

// rho(s), rho'(s) — ceres loss_function.cc
// log(1 + x), x >= 0, as the Cauchy loss needs it.  fp64: log of the rounded sum, what Ceres computes.  fp32: 1 + x rounds
// x away below 6e-8 and v_log_f32 has only absolute accuracy near 1 -- residuals under ~1e-3 would contribute nothing
// (or noise) to the cost, and the solve would stop on "cost = 0" short of the fp64 pose (scripts/soak.py found a
// zero-residual problem ending 1.1e-4 rad away).  Below 2^-7 the series x (1 - x/2 + x^2/3) is used instead: relative
// error < 2e-7 there, and the hardware log2 is relatively accurate above.
template <typename T> __device__ __forceinline__ T t_log1p_of(T x, T sum);
template <> __device__ __forceinline__ double t_log1p_of<double>(double, double sum) { return t_log<double>(sum); }
template <> __device__ __forceinline__ float t_log1p_of<float>(float x, float sum) {
  const float series = x * __builtin_fmaf(x, __builtin_fmaf(x, 0.33333334f, -0.5f), 1.0f);
  return x < 0.0078125f ? series : t_log<float>(sum);
}

template <typename T>
__device__ __forceinline__ void loss_eval(int kind, T a, T inv_b, T s, T &rho, T &w) {
  if (kind == 1) {  // Cauchy
    const T b = a * a;
    const T x = s * inv_b;
    const T sum = x + T(1);
    w = t_rcp<T>(sum);
    rho = b * t_log1p_of<T>(x, sum);
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

// raw 1x6 row of one point from its sample gradient
template <typename T, typename PS>
__device__ __forceinline__ void jacobian_row(const ProblemDesc &pd, const PS &ps,
                                             const Proj<T> &pr, T x, T y, T z, T Fu, T Fv, T J[6]) {
  const T gx = Fu * pr.fxz;
  const T gy = Fv * pr.fyz;
  const T gz = -t_fma<T>(gx, pr.bx, gy * pr.by) * pr.iz;
  if (PoseAcc<T>::unit_q(ps)) {
    // d b / d delta = -2 [R a]x  =>  J_delta = 2 (R a) x g
    J[0] = T(2) * t_fma<T>(pr.cy_, gz, -(pr.cz_ * gy));
    J[1] = T(2) * t_fma<T>(pr.cz_, gx, -(pr.cx_ * gz));
    J[2] = T(2) * t_fma<T>(pr.cx_, gy, -(pr.cy_ * gx));
  } else {
    // (the empty asm keeps this uniform branch a branch: left alone, the compiler computes the 30 extra FMAs of the
    // general path for every point and selects afterwards)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T *G = PoseAcc<T>::G(ps) + 9 * j;
      const T d0 = t_fma<T>(G[2], z, t_fma<T>(G[1], y, G[0] * x));
      const T d1 = t_fma<T>(G[5], z, t_fma<T>(G[4], y, G[3] * x));
      const T d2 = t_fma<T>(G[8], z, t_fma<T>(G[7], y, G[6] * x));
      J[j] = t_fma<T>(gz, d2, t_fma<T>(gy, d1, gx * d0));
    }
  }
  J[3] = gx; J[4] = gy; J[5] = gz;
}

// ------------------------------------------------------------------------------------------------
// wavefront reduction of 32 values per lane: a transposing butterfly.
//
// Each step pairs lanes, lets the pair split its values in two halves, and adds the partner's
// copy of the kept half -- so the number of live values halves while the number of lanes summed
// doubles: 16+8+4+2+1 exchanges instead of 32*6.  The pairings are the ones the hardware offers
// cheaply: lane-swap instructions across half-waves and 16-lane rows (L ^ 32, L ^ 16), DPP inside a
// row (row_mirror = xor 15, row_half_mirror = xor 7, quad_perm = xor 2 and xor 1).  The fp32 and the
// fp64 forms below differ in the order of the levels and hence in the slot a lane ends up with.

constexpr int kDppRowMirror = 0x140, kDppRowHalfMirror = 0x141, kDppQuadXor2 = 0x4E, kDppQuadXor1 = 0xB1;

template <int CTRL>
__device__ __forceinline__ float lane_xchg(float x) {
  // (bound_ctrl: every lane of these patterns has a source lane, so no "old" value is needed -- and none is materialised)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double lane_xchg(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
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
// fp32 wavefront reduction with write-masked DPP adds and lane-swap instructions (gfx950).
//
// Same transposing idea as wave_reduce32, but the "keep my half, add the partner's copy" step is
// two instructions and no select:
//   t = lo + dpp(lo)                       all lanes
//   t = hi + dpp(hi)   bank_mask / rows    only the lanes that keep the upper half
// row_mirror pairs lane L with L^15 and the banks 2,3 (L&8) take the upper half; row_half_mirror
// pairs L with L^7, banks 1,3 (L&4).  The cross-row steps use v_permlane16_swap / v_permlane32_swap:
// swapping the odd rows (upper half-wave) of `lo` with the even rows (lower half-wave) of `hi` and
// adding the two results is exactly the transposing step for L^16 (L^32).  Two plain quad_perm
// adds finish the four lanes of a quad.  64 VALU instructions for 32 slots.
// On return lanes with (L & 3) == 0 hold slots  j + 2*b5 + 4*b4 + 8*b2 + 16*b3  in v[j], j = 0, 1.
//
// The DPP adds are inline asm: hipcc cannot see their read-after-write wait states, so every block
// starts with `s_nop 1` (VALU write -> DPP read of the same VGPR needs two wait states).

#define EA_DPP_FULL(o, a, CTRL) "v_add_f32_dpp %" #o ", %" #a ", %" #a " " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define EA_DPP_MASK(o, a, CTRL, BM) "v_add_f32_dpp %" #o ", %" #a ", %" #a " " CTRL " row_mask:0xf bank_mask:" BM "\n\t"
#define EA_DPP_LEVEL8(CTRL, BM)                                                                              \
  "s_nop 1\n\t" EA_DPP_FULL(0, 8, CTRL) EA_DPP_FULL(1, 9, CTRL) EA_DPP_FULL(2, 10, CTRL) EA_DPP_FULL(3, 11, CTRL) \
      EA_DPP_FULL(4, 12, CTRL) EA_DPP_FULL(5, 13, CTRL) EA_DPP_FULL(6, 14, CTRL) EA_DPP_FULL(7, 15, CTRL)         \
          EA_DPP_MASK(0, 16, CTRL, BM) EA_DPP_MASK(1, 17, CTRL, BM) EA_DPP_MASK(2, 18, CTRL, BM)                 \
              EA_DPP_MASK(3, 19, CTRL, BM) EA_DPP_MASK(4, 20, CTRL, BM) EA_DPP_MASK(5, 21, CTRL, BM)             \
                  EA_DPP_MASK(6, 22, CTRL, BM) EA_DPP_MASK(7, 23, CTRL, BM)

// o[i] = (lanes selected by BM ? hi[i] + partner(hi[i]) : lo[i] + partner(lo[i])), 8 pairs
#define EA_DPP_CALL8(CTRL, BM, o, lo, hi)                                                                     \
  asm volatile(EA_DPP_LEVEL8(CTRL, BM)                                                                         \
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) \
               : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]), "v"(lo[7]),  \
                 "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]))

// v_permlane16_swap / v_permlane32_swap through inline asm (with ROCm 7.2's hipcc the second result of
// __builtin_amdgcn_permlane*_swap aliases the first).  Both operands are swapped in place; the
// leading s_nop covers the VALU-write -> permlane-swap-read wait states hipcc cannot see inside asm.
__device__ __forceinline__ void swap16_x4(float (&lo)[4], float (&hi)[4]) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
               "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 0"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}
__device__ __forceinline__ void swap32_x2(float (&lo)[2], float (&hi)[2]) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 0"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));
}

// levels C and D of the fp32 reduction: 8 values -> 2 (see wave_reduce32_f32)
__device__ __forceinline__ void swap_levels(const float (&b)[8], float (&c)[4], float (&d)[2]) {
  float lo4[4] = {b[0], b[1], b[2], b[3]}, hi4[4] = {b[4], b[5], b[6], b[7]};
  swap16_x4(lo4, hi4);
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = lo4[i] + hi4[i];
  float lo2[2] = {c[0], c[1]}, hi2[2] = {c[2], c[3]};
  swap32_x2(lo2, hi2);
#pragma unroll
  for (int i = 0; i < 2; ++i) d[i] = lo2[i] + hi2[i];
}

__device__ __forceinline__ int masked_slot(int lane, int j) {
  return j + ((lane & 32) >> 4) + ((lane & 16) >> 2) + ((lane & 4) << 1) + ((lane & 8) << 1);
}

__device__ __forceinline__ void wave_reduce32_f32(float (&v)[32]) {
  float a[16], b[8];
  {
    float *o = a, *lo = v, *hi = v + 16;
    EA_DPP_CALL8("row_mirror", "0xc", o, lo, hi);
  }
  {
    float *o = a + 8, *lo = v + 8, *hi = v + 24;
    EA_DPP_CALL8("row_mirror", "0xc", o, lo, hi);
  }
  {
    float *o = b, *lo = a, *hi = a + 8;
    EA_DPP_CALL8("row_half_mirror", "0xa", o, lo, hi);
  }
  float c[4], d[2];
  swap_levels(b, c, d);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    d[i] += lane_xchg<kDppQuadXor1>(d[i]);
    d[i] += lane_xchg<kDppQuadXor2>(d[i]);
  }
  v[0] = d[0];
  v[1] = d[1];
}

// fp64 wavefront reduction with lane-swap instructions for the two widest levels.
//
// v_permlane32_swap exchanges lanes 32..63 of one register with lanes 0..31 of another: applied to slot i and slot
// i + 16 (both dwords of the double), the sum of the two registers is exactly the transposing step for the pairing
// L ^ 32 -- lower half-wave keeps slot i, upper half-wave slot i + 16 -- in three instructions per pair of slots and
// with no select (the DPP form costs seven).  v_permlane16_swap does the same between odd and even 16-lane rows
// (pairing L ^ 16).  The three levels inside a row stay on DPP (row_mirror, row_half_mirror, quad_perm), the last
// pairing (L ^ 1) is a plain add.  ~125 instructions instead of ~220.
// On return lanes with even L hold in v[0] the wave total of slot  16 b5 + 8 b4 + 4 b3 + 2 b2 + b1  (b_k = bit k of L).
// four pairs of doubles per asm statement: one pair of wait-state s_nops (VALU write -> permlane-swap read, swap
// write -> VALU read; hipcc cannot see inside the asm) per eight swaps instead of per two
#define EA_SWAP8(OP)                                                                                        \
  "s_nop 1\n\t" OP " %0, %8\n\t" OP " %1, %9\n\t" OP " %2, %10\n\t" OP " %3, %11\n\t" OP " %4, %12\n\t" \
  OP " %5, %13\n\t" OP " %6, %14\n\t" OP " %7, %15\n\ts_nop 0"
template <int W>  // W = 32: v_permlane32_swap, 16: v_permlane16_swap; a[k] <-> b[k] for k = 0..3 (both dwords)
__device__ __forceinline__ void swap_f64x4(double *a, double *b) {
  unsigned al[4], ah[4], bl[4], bh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a[k]), ub = __builtin_bit_cast(unsigned long long, b[k]);
    al[k] = (unsigned)ua; ah[k] = (unsigned)(ua >> 32); bl[k] = (unsigned)ub; bh[k] = (unsigned)(ub >> 32);
  }
  if constexpr (W == 32)
    asm volatile(EA_SWAP8("v_permlane32_swap_b32")
                 : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]),
                   "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3]));
  else
    asm volatile(EA_SWAP8("v_permlane16_swap_b32")
                 : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]),
                   "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3]));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = __builtin_bit_cast(double, ((unsigned long long)ah[k] << 32) | al[k]);
    b[k] = __builtin_bit_cast(double, ((unsigned long long)bh[k] << 32) | bl[k]);
  }
}

__device__ __forceinline__ int swap_slot_f64(int lane) {
  return ((lane & 32) >> 1) | ((lane & 16) >> 1) | ((lane & 8) >> 1) | ((lane & 4) >> 1) | ((lane & 2) >> 1);
}

__device__ __forceinline__ void wave_reduce32_f64(double (&v)[32], int lane) {
#pragma unroll
  for (int i = 0; i < 16; i += 4) {  // L ^ 32
    swap_f64x4<32>(v + i, v + i + 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[i + k] += v[i + k + 16];
  }
#pragma unroll
  for (int i = 0; i < 8; i += 4) {  // L ^ 16
    swap_f64x4<16>(v + i, v + i + 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[i + k] += v[i + k + 8];
  }
  // inside a 16-lane row: L ^ 15 (bit 3 picks the half kept), L ^ 7 (bit 2), L ^ 2 (bit 1)
  {
    const bool upper = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double send = upper ? v[i] : v[i + 4];
      const double keep = upper ? v[i + 4] : v[i];
      v[i] = keep + lane_xchg<kDppRowMirror>(send);
    }
  }
  {
    const bool upper = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double send = upper ? v[i] : v[i + 2];
      const double keep = upper ? v[i + 2] : v[i];
      v[i] = keep + lane_xchg<kDppRowHalfMirror>(send);
    }
  }
  {
    const bool upper = (lane & 2) != 0;
    const double send = upper ? v[0] : v[1];
    const double keep = upper ? v[1] : v[0];
    v[0] = keep + lane_xchg<kDppQuadXor2>(send);
  }
  v[0] += lane_xchg<kDppQuadXor1>(v[0]);
}

#ifndef EA_TU_VARIANT
// self-test of the cross-lane primitives: one wave, lane L loads in[s*64+L] for slot s; dumps the
// stages of wave_reduce32_f32 (a: 16x64, b: 8x64, c: 4x64, d: 2x64) and the fp64 butterfly result
__global__ void ea_selftest_reduce_kernel(const float *in, float *stage_a, float *stage_b, float *stage_c,
                                          float *stage_d, double *out_f32 /*32*/, double *out_f64 /*32*/) {
  const int lane = threadIdx.x & 63;
  float v[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) v[s] = in[s * 64 + lane];
  double w[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) w[s] = (double)v[s];
  float a[16], b[8];
  { float *o = a, *lo = v, *hi = v + 16; EA_DPP_CALL8("row_mirror", "0xc", o, lo, hi); }
  { float *o = a + 8, *lo = v + 8, *hi = v + 24; EA_DPP_CALL8("row_mirror", "0xc", o, lo, hi); }
  { float *o = b, *lo = a, *hi = a + 8; EA_DPP_CALL8("row_half_mirror", "0xa", o, lo, hi); }
  float c[4], d[2];
  swap_levels(b, c, d);
#pragma unroll
  for (int i = 0; i < 16; ++i) stage_a[i * 64 + lane] = a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_b[i * 64 + lane] = b[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) stage_c[i * 64 + lane] = c[i];
#pragma unroll
  for (int i = 0; i < 2; ++i) stage_d[i * 64 + lane] = d[i];
  wave_reduce32_f32(v);
  if ((lane & 3) == 0) {
    out_f32[masked_slot(lane, 0)] = (double)v[0];
    out_f32[masked_slot(lane, 1)] = (double)v[1];
  }
  wave_reduce32_f64(w, lane);
  if ((lane & 1) == 0) out_f64[swap_slot_f64(lane)] = w[0];
}
#endif  // !EA_TU_VARIANT

template <typename T> using GPtr = const T __attribute__((address_space(1))) *;

// four consecutive texels through a global-address-space pointer (merged into dwordx4 loads)
template <typename T>
__device__ __forceinline__ Row4<T> load_row4(GPtr<T> p) {
  Row4<T> r;
  r.p0 = p[0]; r.p1 = p[1]; r.p2 = p[2]; r.p3 = p[3];
  return r;
}

// ------------------------------------------------------------------------------------------------
// raw-buffer addressing (gfx950 buffer instructions): a lane's address is ONE 32-bit byte offset and the four
// stencil rows of a point differ only in the instruction's SGPR offset (l * pitch bytes) -- no 64-bit address
// arithmetic per row (ten v_lshl_add_u64 per point in the flat-address form).  Requires the padded image to be
// smaller than 2 GiB (checked on the host, which falls back to the flat-address kernels otherwise).

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_raw_buffer(const void *base, unsigned bytes) {
  // gfx950 raw buffer: stride 0, no swizzle, DATA_FORMAT = 32 (dword 3 = 0x00020000)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

template <typename T> __device__ __forceinline__ T buf_load_elem(__amdgpu_buffer_rsrc_t r, int voff);
template <> __device__ __forceinline__ float buf_load_elem<float>(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
template <> __device__ __forceinline__ double buf_load_elem<double>(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}

template <typename T> __device__ __forceinline__ Row4<T> buf_load_row4(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ Row4<float> buf_load_row4<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  // (whole-vector bit casts: __builtin_bit_cast of a single element of the loaded vector, `v.x`, comes out of
  // ROCm 7.2's hipcc as four reads of the same dword)
  const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  Row4<float> o;
  o.p0 = v[0]; o.p1 = v[1]; o.p2 = v[2]; o.p3 = v[3];
  return o;
}
template <> __device__ __forceinline__ Row4<double> buf_load_row4<double>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const f64x2_t a = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  const f64x2_t b = __builtin_bit_cast(f64x2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, soff, 0));
  Row4<double> o;
  o.p0 = a[0]; o.p1 = a[1]; o.p2 = b[0]; o.p3 = b[1];
  return o;
}

// A stencil row of a T-typed kernel out of an image stored as IT (IT = T, or float under a double kernel: the fp32-stored
// distance transform of an fp64 problem -- ONE 16-byte load per row and four exact conversions instead of two loads).
template <typename T, typename IT> struct RowLoad {
  static __device__ __forceinline__ Row4<T> buf(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return buf_load_row4<T>(r, voff, soff); }
  static __device__ __forceinline__ Row4<T> flat(GPtr<IT> p) { return load_row4<T>(p); }
};
template <> struct RowLoad<double, float> {
  static __device__ __forceinline__ Row4<double> buf(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const Row4<float> f = buf_load_row4<float>(r, voff, soff);
    Row4<double> o;
    o.p0 = (double)f.p0; o.p1 = (double)f.p1; o.p2 = (double)f.p2; o.p3 = (double)f.p3;
    return o;
  }
  static __device__ __forceinline__ Row4<double> flat(GPtr<float> p) {
    const Row4<float> f = load_row4<float>(p);
    Row4<double> o;
    o.p0 = (double)f.p0; o.p1 = (double)f.p1; o.p2 = (double)f.p2; o.p3 = (double)f.p3;
    return o;
  }
};
template <typename T, bool IMG32> struct ImgOf { typedef T type; };
template <> struct ImgOf<double, true> { typedef float type; };

// ------------------------------------------------------------------------------------------------
// fused evaluation kernel: residual + Jacobian + loss + JtJ/Jtr/cost partials, one row per workgroup
//
// grid = (8 * ceil(chunks/8), problems).  Workgroup (c, p) owns points [c*chunk, (c+1)*chunk) of
// problem p, chunk = NT*PPT: every lane takes PPT points (strided by NT so loads coalesce), one
// wavefront butterfly per wave and one cross-wave fold per workgroup -> one partial row.

constexpr int kMaxWaves = 16;
constexpr int kRedBytes = kMaxWaves * kAccSlots * 8;  // cross-wave scratch: up to 16 waves x 32 doubles
constexpr int kHdrBytes = kRedBytes + kMaxWaves * 16;  // + bbox words, keeps the tile 16-byte aligned

// One workgroup's share of an evaluation: NT lanes x PPT points -> the workgroup's partial row.
// X/Y/Z hold the lane's points (lanes past `count` carry a copy of the chunk's last point); the return value is
// slot `my_slot` of the row (0 when my_slot < 0).  `ps` may live in global memory (scalar loads) or LDS.
template <typename T, int PPT, int MODE, int NT, bool VAR, bool BUF, bool IMG32, typename PS>
__device__ __forceinline__ double fused_chunk(const ProblemDesc &pd, const PS &ps, const T (&X)[PPT],
                                              const T (&Y)[PPT], const T (&Z)[PPT], int count, double *s_red,
                                              int *s_box, T *s_tile, int lds_texels, int my_slot) {
  // MODE 0: stencil rows from L2.  1: DT footprint staged in LDS.  2: as 0, and an fp32 kernel widens a lane's sums to
  // fp64 before the wavefront butterfly ("wide_accumulate": everything above a lane's <= PPT products is fp64).
  constexpr bool USE_LDS = MODE == 1;
  constexpr bool WIDE = MODE == 2 && sizeof(T) == 4;
  static_assert(!IMG32 || (std::is_same<T, double>::value && MODE == 0 && !VAR), "fp32-stored image: plain fp64 kernels on the L2 path");
  typedef typename ImgOf<T, IMG32>::type IT;  // element type of the image in HBM
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int pitch = pd.pitch;
  const void *img_base = IMG32 ? pd.dt32 : pd.dt;
  const GPtr<IT> gimg = (GPtr<IT>)(static_cast<const IT *>(img_base) + (size_t)kImagePad * (size_t)pitch + kImagePad);
  const __amdgpu_buffer_rsrc_t rimg =
      make_raw_buffer(img_base, BUF ? (unsigned)pitch * (unsigned)(pd.H + 2 * kImagePad) * (unsigned)sizeof(IT) : 0u);
  const int loss_kind = pd.loss_kind;
  const T loss_a = Uni<T>::loss_a(pd), loss_inv_b = Uni<T>::loss_inv_b(pd);

  T acc[28];
  int n_bad = 0;
  // ---- phase 1: warp + projection.  Lanes past the end of the chunk and lanes whose functor fails are moved
  // to a harmless sample; both get weight 0 below, so the arithmetic needs no divergent branch.
  using ProjT = typename std::conditional<VAR, ProjV<T>, Proj<T>>::type;
  ProjT pr[PPT];
  bool valid[PPT];
  int bb_u0 = 0x7fffffff, bb_u1 = -0x7fffffff, bb_v0 = 0x7fffffff, bb_v1 = -0x7fffffff;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const bool inb = tid + k * NT < count;
    if constexpr (VAR) project_point_var<T>(pd, ps, X[k], Y[k], Z[k], pr[k]);
    else project_point<T>(pd, ps, X[k], Y[k], Z[k], pr[k]);
    valid[k] = inb && pr[k].state == 1;
    n_bad += (inb && pr[k].state == 2) ? 1 : 0;
    if (__any(!valid[k]) && !valid[k]) {  // (uniform test first: the common wavefront has nothing to fix up)
      pr[k].iu = 0; pr[k].iv = 0; pr[k].fu = T(0); pr[k].fv = T(0);
      pr[k].iz = T(1);
      if constexpr (VAR) {
        pr[k].x = T(0); pr[k].y = T(0);
        pr[k].xdx = T(1); pr[k].xdy = T(0); pr[k].ydx = T(0); pr[k].ydy = T(1);
      } else {
        pr[k].bx = T(0); pr[k].by = T(0);
        pr[k].fxz = T(0); pr[k].fyz = T(0);
      }
    }
    if (USE_LDS && valid[k]) {
      bb_u0 = min(bb_u0, pr[k].iu); bb_u1 = max(bb_u1, pr[k].iu);
      bb_v0 = min(bb_v0, pr[k].iv); bb_v1 = max(bb_v1, pr[k].iv);
    }
  }

  // ---- phase 2: footprint of the chunk, staged through LDS when it fits
  bool in_lds = false;
  int u0 = 0, v0 = 0, tw = 0;
  if (USE_LDS) {
    bb_u0 = wave_min_i32(bb_u0); bb_u1 = wave_max_i32(bb_u1);
    bb_v0 = wave_min_i32(bb_v0); bb_v1 = wave_max_i32(bb_v1);
    if (lane == 0) {
      s_box[4 * wave + 0] = bb_u0; s_box[4 * wave + 1] = bb_u1;
      s_box[4 * wave + 2] = bb_v0; s_box[4 * wave + 3] = bb_v1;
    }
    __syncthreads();
    int U0 = s_box[0], U1 = s_box[1], V0 = s_box[2], V1 = s_box[3];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) {
      U0 = min(U0, s_box[4 * w + 0]); U1 = max(U1, s_box[4 * w + 1]);
      V0 = min(V0, s_box[4 * w + 2]); V1 = max(V1, s_box[4 * w + 3]);
    }
    if (U1 >= U0) {
      u0 = min(U0, 0) - 1; v0 = min(V0, 0) - 1;  // weight-0 lanes sample texel (0,0): keep it inside
      U1 = max(U1, 0); V1 = max(V1, 0);
      tw = U1 - (u0 + 1) + 4;
      const int th = V1 - (v0 + 1) + 4;
      const long long area = (long long)tw * (long long)th;
      if (area <= (long long)lds_texels) {
        in_lds = true;
        const GPtr<IT> img = gimg + ((ptrdiff_t)v0 * pitch + u0);
        const float inv_tw = 1.0f / (float)tw;
        for (int idx = tid; idx < (int)area; idx += NT) {
          int row = (int)((float)idx * inv_tw);
          int col = idx - row * tw;
          if (col < 0) { row -= 1; col += tw; }
          if (col >= tw) { row += 1; col -= tw; }
          s_tile[idx] = img[(ptrdiff_t)row * pitch + col];
        }
        __syncthreads();
      }
    }
  }

  EA_STAMP(3);  // projected
  // ---- phase 3: sample, Jacobian, weights, accumulate
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    T f, Fu, Fv;
    if (USE_LDS && in_lds) {
      const T *base = s_tile + (pr[k].iv - 1 - v0) * tw + (pr[k].iu - 1 - u0);
      const int stride = tw;
      bicubic<T>(pr[k].fu, pr[k].fv,
                 [&](int l) { return *reinterpret_cast<const Row4<T> *>(base + l * stride); }, f, Fu, Fv);
    } else if constexpr (BUF) {
      // byte offset of texel (iv - 1, iu - 1) in the padded image; iv, iu >= -2 keeps it non-negative
      const int voff = ((pr[k].iv + (kImagePad - 1)) * pitch + (pr[k].iu + (kImagePad - 1))) * (int)sizeof(IT);
      bicubic<T>(pr[k].fu, pr[k].fv,
                 [&](int l) { return RowLoad<T, IT>::buf(rimg, voff, l * pitch * (int)sizeof(IT)); }, f, Fu, Fv);
    } else {
      const GPtr<IT> base = gimg + ((ptrdiff_t)(pr[k].iv - 1) * pitch + (pr[k].iu - 1));
      bicubic<T>(pr[k].fu, pr[k].fv,
                 [&](int l) { return RowLoad<T, IT>::flat(base + (ptrdiff_t)l * pitch); }, f, Fu, Fv);
    }
    T J[6];
    if constexpr (VAR) jacobian_row_var<T>(pd, ps, pr[k], Fu, Fv, J);
    else jacobian_row<T>(pd, ps, pr[k], X[k], Y[k], Z[k], Fu, Fv, J);
    T rho, w;
    loss_eval<T>(loss_kind, loss_a, loss_inv_b, f * f, rho, w);
    w = valid[k] ? w : T(0);
    rho = valid[k] ? rho : T(0);
    const T wr = w * f;
    int s = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const T wJa = w * J[a];
#pragma unroll
      for (int b = a; b < 6; ++b) { acc[s] = k == 0 ? wJa * J[b] : t_fma<T>(wJa, J[b], acc[s]); ++s; }
      acc[kAccJtr + a] = k == 0 ? J[a] * wr : t_fma<T>(J[a], wr, acc[kAccJtr + a]);
    }
    acc[kAccCost] = k == 0 ? T(0.5) * rho : t_fma<T>(T(0.5), rho, acc[kAccCost]);
  }

  // ---- phase 4: wavefront reduction in the kernel's arithmetic type (a lane's accumulators and
  // a wavefront's 64-lane sums are T; everything above a wavefront is fp64), then the
  // fixed-order cross-wave sum
  T v[32];
#pragma unroll
  for (int i = 0; i < 28; ++i) v[i] = acc[i];
  v[28] = (T)n_bad; v[29] = T(0); v[30] = T(0); v[31] = T(0);
#ifdef EA_STAMPS
  asm volatile("" ::"v"(v[0]), "v"(v[27]));
#endif
  EA_STAMP(4);  // sampled + accumulated
  if (USE_LDS) __syncthreads();  // tile readers done before the scratch rows are written
  if constexpr (WIDE) {
    double w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = (double)v[i];
    wave_reduce32_f64(w, lane);
    if ((lane & 1) == 0) s_red[wave * kAccSlots + swap_slot_f64(lane)] = w[0];
  } else if constexpr (sizeof(T) == 4) {
    wave_reduce32_f32(v);
    if ((lane & 3) == 0) {
      s_red[wave * kAccSlots + masked_slot(lane, 0)] = (double)v[0];
      s_red[wave * kAccSlots + masked_slot(lane, 1)] = (double)v[1];
    }
  } else {
    wave_reduce32_f64(v, lane);
    if ((lane & 1) == 0) s_red[wave * kAccSlots + swap_slot_f64(lane)] = v[0];
  }
  EA_STAMP(5);  // wave reduced
  __syncthreads();
  EA_STAMP(6);  // all waves of the workgroup arrived
  double sum = 0.0;
  if (my_slot >= 0) {
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) sum += s_red[w * kAccSlots + my_slot];
  }
  return sum;
}

// a value every lane of the wavefront holds (read from LDS, say) as a scalar-register operand
template <typename T> __device__ __forceinline__ T uniform_scalar(T v);
template <> __device__ __forceinline__ float uniform_scalar<float>(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <> __device__ __forceinline__ double uniform_scalar<double>(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// NT = workgroup size (256 or 1024).  1024 = one workgroup per CU: four times fewer partial rows
// to fold afterwards at the same points-per-lane latency.
template <typename T, int PPT, int MODE, int NT, bool VAR, bool BUF, bool IMG32>
__device__ __forceinline__ void eval_fused_body(
    const void *__restrict__ x0, const void *__restrict__ y0, const void *__restrict__ z0, int n0,
    int shape, int chunks_per_xcd,
    const ProblemDesc *__restrict__ probs, const PoseState *__restrict__ poses,
    double *__restrict__ partials, int lds_texels) {
  const int chunk = shape & 0xffff, xcd_remap = (shape >> 16) & 1, terms_are_groups = (shape >> 17) & 1;
  extern __shared__ __align__(16) unsigned char smem[];
  double *s_red = reinterpret_cast<double *>(smem);
  int *s_box = reinterpret_cast<int *>(smem + kRedBytes);
  T *s_tile = reinterpret_cast<T *>(smem + kHdrBytes);

  // XCD-aware chunk assignment: workgroups are dealt round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of chunks (neighbouring chunks read neighbouring image rows -> one L2).
  EA_STAMP(0);
  const int bx = blockIdx.x;
  const int c = xcd_remap ? (bx & 7) * chunks_per_xcd + (bx >> 3) : bx;
  const long long start = (long long)c * chunk;
  const int tid = threadIdx.x;
  T X[PPT], Y[PPT], Z[PPT];
  // Problem 0 (every single-problem launch: C2, an LM iteration): its point arrays and count came with the wave, so the
  // coalesced point loads go out now and travel while the descriptor and the pose are fetched -- one dependent memory
  // round trip less in the chain descriptor -> points -> stencil rows that bounds a small launch.
  const bool early = BUF && blockIdx.y == 0 && n0 > 0;  // (uniform)
  if constexpr (BUF) {
    if (early) {
      if (start >= n0) return;
      const int count0 = min(chunk, (int)(n0 - start));
      const __amdgpu_buffer_rsrc_t rx = make_raw_buffer(static_cast<const T *>(x0) + start, (unsigned)count0 * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t ry = make_raw_buffer(static_cast<const T *>(y0) + start, (unsigned)count0 * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t rz = make_raw_buffer(static_cast<const T *>(z0) + start, (unsigned)count0 * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int poff = min(tid + k * NT, count0 - 1) * (int)sizeof(T);
        X[k] = buf_load_elem<T>(rx, poff); Y[k] = buf_load_elem<T>(ry, poff); Z[k] = buf_load_elem<T>(rz, poff);
      }
    }
  }
  // descriptor and pose by value: every scalar load is issued here, behind one wait, instead of a
  // chain of dependent loads at the points of use
  const ProblemDesc pd = probs[blockIdx.y];
  // one pose per group of terms; when every problem is a single residual family the group index is
  // the term index and the pose load does not have to wait for the descriptor
  // The pose is fetched SPECULATIVELY from the slot of this term's own index -- the right one whenever every problem is a
  // single residual family (terms_are_groups), and then independent of the descriptor -- into scalars, beside the
  // descriptor.  Every uniform the common path reads is named in front of the early exit: descriptor and pose go out as
  // ONE batch of scalar loads behind one wait.  Left alone the compiler fetches what the exit test needs, then the
  // pose's flag, then the rest: three dependent round trips at the head of every workgroup.
  PoseLite<T> ps;
  int active;
  {
    const PoseState *psp = poses + blockIdx.y;
    const T *R_ = Uni<T>::R(*psp), *t_ = Uni<T>::t(*psp);
#pragma unroll
    for (int i = 0; i < 9; ++i) ps.R[i] = R_[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.t[i] = t_[i];
    ps.unit_q = psp->unit_q;
    ps.full = psp;
    active = psp->active;
    asm volatile("" ::"s"(pd.x), "s"(pd.y), "s"(pd.z), "s"(IMG32 ? pd.dt32 : pd.dt), "s"(pd.n), "s"(pd.W), "s"(pd.H), "s"(pd.pitch),
                 "s"(Uni<T>::fx(pd)), "s"(Uni<T>::fy(pd)), "s"(Uni<T>::cx(pd)), "s"(Uni<T>::cy(pd)),
                 "s"(Uni<T>::loss_a(pd)), "s"(Uni<T>::loss_inv_b(pd)), "s"(Uni<T>::z_guard(pd)), "s"(Uni<T>::z_eps(pd)),
                 "s"(pd.loss_kind), "s"(pd.tile_begin), "s"(pd.variant), "s"(pd.group));
    asm volatile("" : "+s"(active), "+s"(ps.unit_q), "+s"(ps.R[0]), "+s"(ps.R[1]), "+s"(ps.R[2]), "+s"(ps.R[3]), "+s"(ps.R[4]),
                      "+s"(ps.R[5]), "+s"(ps.R[6]), "+s"(ps.R[7]), "+s"(ps.R[8]), "+s"(ps.t[0]), "+s"(ps.t[1]), "+s"(ps.t[2]));
    if (!terms_are_groups && pd.group != (int)blockIdx.y) {  // (uniform, rare: a term that shares another term's pose)
      psp = poses + pd.group;
      R_ = Uni<T>::R(*psp); t_ = Uni<T>::t(*psp);
#pragma unroll
      for (int i = 0; i < 9; ++i) ps.R[i] = R_[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) ps.t[i] = t_[i];
      ps.unit_q = psp->unit_q;
      ps.full = psp;
      active = psp->active;
    }
  }
  if (start >= pd.n || !active) return;
  const int count = min(chunk, (int)(pd.n - start));
  EA_STAMP(1);  // descriptor + pose.active arrived

  const GPtr<T> px = (GPtr<T>)(static_cast<const T *>(pd.x) + start);
  const GPtr<T> py = (GPtr<T>)(static_cast<const T *>(pd.y) + start);
  const GPtr<T> pz = (GPtr<T>)(static_cast<const T *>(pd.z) + start);
  // coalesced point loads; lanes past the end of the chunk re-read its last point
  if constexpr (BUF) {
    if (!early) {
      // the chunk's points as three raw buffers: one 32-bit offset per lane serves all three loads
      const __amdgpu_buffer_rsrc_t rx = make_raw_buffer((const T *)px, (unsigned)count * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t ry = make_raw_buffer((const T *)py, (unsigned)count * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t rz = make_raw_buffer((const T *)pz, (unsigned)count * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int poff = min(tid + k * NT, count - 1) * (int)sizeof(T);
        X[k] = buf_load_elem<T>(rx, poff); Y[k] = buf_load_elem<T>(ry, poff); Z[k] = buf_load_elem<T>(rz, poff);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int j = tid + k * NT;
      const int jj = min(j, count - 1);
      X[k] = px[jj]; Y[k] = py[jj]; Z[k] = pz[jj];
    }
  }
#ifdef EA_STAMPS
  asm volatile("" ::"v"(X[0]), "v"(Y[0]), "v"(Z[0]));
#endif
  EA_STAMP(2);  // points arrived
  const double sum = fused_chunk<T, PPT, MODE, NT, VAR, BUF, IMG32, PoseLite<T>>(pd, ps, X, Y, Z, count, s_red, s_box, s_tile, lds_texels,
                                                           tid < kAccSlots ? tid : -1);
  if (tid < kAccSlots) partials[(size_t)(pd.tile_begin + c) * kAccSlots + tid] = sum;
#ifdef EA_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  EA_STAMP(7);  // row stored
}

template <typename T, int PPT, int MODE, int NT, bool VAR, bool BUF, bool IMG32 = false>
__global__ __launch_bounds__(NT) void ea_eval_fused_kernel(
    // the first 16 dwords of the argument segment arrive in SGPRs with the wave (kernarg preload): what problem 0's point
    // loads need sits there, so that they can be issued at once, beside the descriptor fetch instead of behind it
    // (14 dwords fit beside the argument pointer: three pointers, the count, the launch shape packed into one word --
    // chunk | xcd_remap << 16 | terms_are_groups << 17 --, chunks per XCD, and the descriptor and pose tables)
    const void *__restrict__ x0, const void *__restrict__ y0, const void *__restrict__ z0, int n0,
    int shape, int chunks_per_xcd,
    const ProblemDesc *__restrict__ probs, const PoseState *__restrict__ poses,
    double *__restrict__ partials, int lds_texels) {
  eval_fused_body<T, PPT, MODE, NT, VAR, BUF, IMG32>(x0, y0, z0, n0, shape, chunks_per_xcd, probs, poses, partials, lds_texels);
}

// The same kernel under a second name, for the launches of ea_batch_eval_poses (grid y = G poses x terms over a descriptor
// table replicated G times): rocprofv3 --kernel-trace --stats then keeps the pose-batched launches -- the dominant kernel
// of bench.py's timed region -- apart from the one-pose launches of the solves and of ea_batch_eval in the same process,
// and its average duration can be read off the summary.  Body, arguments and instantiations are identical.
template <typename T, int PPT, int MODE, int NT, bool VAR, bool BUF, bool IMG32 = false>
__global__ __launch_bounds__(NT) void ea_eval_poses_kernel(
    const void *__restrict__ x0, const void *__restrict__ y0, const void *__restrict__ z0, int n0,
    int shape, int chunks_per_xcd,
    const ProblemDesc *__restrict__ probs, const PoseState *__restrict__ poses,
    double *__restrict__ partials, int lds_texels) {
  eval_fused_body<T, PPT, MODE, NT, VAR, BUF, IMG32>(x0, y0, z0, n0, shape, chunks_per_xcd, probs, poses, partials, lds_texels);
}

typedef void (*EvalKernelFn)(const void *, const void *, const void *, int, int, int, const ProblemDesc *, const PoseState *, double *, int);
template <int TAG, typename T, int PPT, int MODE, int NT, bool VAR, bool BUF, bool IMG32>
struct EvalKernel { static constexpr EvalKernelFn fn = &ea_eval_fused_kernel<T, PPT, MODE, NT, VAR, BUF, IMG32>; };
template <typename T, int PPT, int MODE, int NT, bool VAR, bool BUF, bool IMG32>
struct EvalKernel<1, T, PPT, MODE, NT, VAR, BUF, IMG32> { static constexpr EvalKernelFn fn = &ea_eval_poses_kernel<T, PPT, MODE, NT, VAR, BUF, IMG32>; };

#ifndef EA_TU_VARIANT
// ------------------------------------------------------------------------------------------------
// per-point outputs (parity / "EAResidue batch Evaluate" view): r[n], J[n*6]

template <typename T>
__global__ __launch_bounds__(kBlockThreads) void ea_eval_points_kernel(
    const ProblemDesc *__restrict__ probs, int problem, const PoseState *__restrict__ poses,
    double *__restrict__ r_out, double *__restrict__ J_out, int corrected) {
  const ProblemDesc &pd = probs[problem];
  const PoseState &ps = poses[pd.group];
  const int i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= pd.n) return;
  const T x = static_cast<const T *>(pd.x)[i], y = static_cast<const T *>(pd.y)[i], z = static_cast<const T *>(pd.z)[i];
  const double nan = __builtin_nan("");
  if (pd.variant) {  // distortion / second-camera functors
    ProjV<T> pv;
    project_point_var<T>(pd, ps, x, y, z, pv);
    if (pv.state != 1) {
      if (r_out) r_out[i] = nan;
      if (J_out)
        for (int a = 0; a < 6; ++a) J_out[(size_t)i * 6 + a] = nan;
      return;
    }
    const int pitchv = pd.pitch;
    const T *basev = static_cast<const T *>(pd.dt) + (size_t)kImagePad * (size_t)pitchv + kImagePad +
                     (ptrdiff_t)(pv.iv - 1) * pitchv + (pv.iu - 1);
    T f, Fu, Fv;
    bicubic<T>(pv.fu, pv.fv, [&](int l) { return *reinterpret_cast<const Row4<T> *>(basev + (ptrdiff_t)l * pitchv); }, f, Fu, Fv);
    T J[6];
    jacobian_row_var<T>(pd, ps, pv, Fu, Fv, J);
    T sc = T(1);
    if (corrected) {
      T rho, w;
      loss_eval<T>(pd.loss_kind, Uni<T>::loss_a(pd), Uni<T>::loss_inv_b(pd), f * f, rho, w);
      sc = t_sqrt<T>(w);
    }
    if (r_out) r_out[i] = (double)(sc * f);
    if (J_out)
      for (int a = 0; a < 6; ++a) J_out[(size_t)i * 6 + a] = (double)(sc * J[a]);
    return;
  }
  Proj<T> pr;
  project_point<T>(pd, ps, x, y, z, pr);
  if (pr.state != 1) {
    if (r_out) r_out[i] = nan;
    if (J_out)
      for (int a = 0; a < 6; ++a) J_out[(size_t)i * 6 + a] = nan;
    return;
  }
  const int pitch = pd.pitch;
  const T *base = static_cast<const T *>(pd.dt) + (size_t)kImagePad * (size_t)pitch + kImagePad +
                  (ptrdiff_t)(pr.iv - 1) * pitch + (pr.iu - 1);
  T f, Fu, Fv;
  bicubic<T>(pr.fu, pr.fv, [&](int l) { return *reinterpret_cast<const Row4<T> *>(base + (ptrdiff_t)l * pitch); }, f, Fu, Fv);
  T J[6];
  jacobian_row<T>(pd, ps, pr, x, y, z, Fu, Fv, J);
  T sc = T(1);
  if (corrected) {
    T rho, w;
    loss_eval<T>(pd.loss_kind, Uni<T>::loss_a(pd), Uni<T>::loss_inv_b(pd), f * f, rho, w);
    sc = t_sqrt<T>(w);
  }
  if (r_out) r_out[i] = (double)(sc * f);
  if (J_out)
    for (int a = 0; a < 6; ++a) J_out[(size_t)i * 6 + a] = (double)(sc * J[a]);
}

// ------------------------------------------------------------------------------------------------
// Materialised mode (the "EAResidue batch Evaluate" view): residual and 1x6 row of EVERY point written out in
// the problem's arithmetic type -- what N calls of AutoDiffCostFunction<EAResidue,1,4,3>::Evaluate followed by the
// parameterisation's 4x3 plus-Jacobian produce (standalone/utils.h:48-92, standalone_edge_align.cpp:271-278), for a caller
// that runs its own solver on the rows.  One point per lane, 256-lane workgroups, the chunk -> XCD mapping and the
// raw-buffer stencil loads of the fused kernel, no reduction.  Unlike the fused mode this one IS bandwidth-bound:
// 3 s bytes in and 7 s bytes out per point plus one pass over the DT image (s = sizeof(T)).
//   LAYOUT 0: J row-major [rows][6] (what Ceres hands its linear solver); the 6 values of a lane are 24 / 48 contiguous
//             bytes, so a wavefront's 64 rows go through an LDS transpose and leave as three stores of 64 x 8 / 16
//             contiguous bytes (STAGED; the direct form -- three 8/16-byte stores per lane at a 24/48-byte stride -- is kept
//             for comparison);
//   LAYOUT 1: J column-major [6][total rows] (six coalesced stores, no staging).
// Rows of a functor that returned false are NaN (and counted): Ceres fails the whole evaluation in that case.
// `corrected`: rows scaled by sqrt(rho'(r^2)) (Ceres' Corrector with alpha = 0, as the fused mode accumulates them).
template <typename T> struct PairOf;
template <> struct PairOf<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct PairOf<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T> using Pair = typename PairOf<T>::type;  // two consecutive elements: one 8- / 16-byte access

#ifndef EA_ROWS_WAVES_F64
#define EA_ROWS_WAVES_F64 4  // (fp64 occupancy target of the rows kernel, wavefronts per SIMD: A/B knob, scripts/archive/ab_rows.sh)
#endif
template <typename T, bool VAR, bool BUF, int LAYOUT, bool STAGED, bool IMG32 = false>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? EA_ROWS_WAVES_F64 : 8, 8)))
void ea_eval_rows_kernel(
    const ProblemDesc *__restrict__ probs, const PoseState *__restrict__ poses, int chunks_per_xcd, int corrected, int nontemporal,
    long long total_rows, T *__restrict__ r_out, T *__restrict__ J_out, unsigned int *__restrict__ n_invalid) {
  constexpr int NT = kBlockThreads;
  const int bx = blockIdx.x;
  const int c = (bx & 7) * chunks_per_xcd + (bx >> 3);
  const ProblemDesc pd = probs[blockIdx.y];
  const long long start = (long long)c * NT;
  if (start >= pd.n) return;  // (uniform)
  const PoseState *psp = poses + pd.group;
  PoseLite<T> ps;
  {
    const T *R_ = Uni<T>::R(*psp), *t_ = Uni<T>::t(*psp);
#pragma unroll
    for (int i = 0; i < 9; ++i) ps.R[i] = R_[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.t[i] = t_[i];
    ps.unit_q = psp->unit_q;
    ps.full = psp;
  }
  const int count = min(NT, (int)(pd.n - start));
  const int tid = threadIdx.x;
  const int j = min(tid, count - 1);  // lanes past the end re-read the chunk's last point and store nothing
  const bool inb = tid < count;
  T X, Y, Z;
  if constexpr (BUF) {
    const __amdgpu_buffer_rsrc_t rx = make_raw_buffer(static_cast<const T *>(pd.x) + start, (unsigned)count * (unsigned)sizeof(T));
    const __amdgpu_buffer_rsrc_t ry = make_raw_buffer(static_cast<const T *>(pd.y) + start, (unsigned)count * (unsigned)sizeof(T));
    const __amdgpu_buffer_rsrc_t rz = make_raw_buffer(static_cast<const T *>(pd.z) + start, (unsigned)count * (unsigned)sizeof(T));
    const int poff = j * (int)sizeof(T);
    X = buf_load_elem<T>(rx, poff); Y = buf_load_elem<T>(ry, poff); Z = buf_load_elem<T>(rz, poff);
  } else {
    X = static_cast<const T *>(pd.x)[start + j]; Y = static_cast<const T *>(pd.y)[start + j]; Z = static_cast<const T *>(pd.z)[start + j];
  }
  const int pitch = pd.pitch;
  T f, Fu, Fv, J[6];
  int state;
  static_assert(!IMG32 || (std::is_same<T, double>::value && !VAR), "fp32-stored image: plain fp64 kernels");
  typedef typename ImgOf<T, IMG32>::type IT;
  const void *img_base = IMG32 ? pd.dt32 : pd.dt;
  auto sample = [&](int iu, int iv, T fu, T fv) {
    if constexpr (BUF) {
      const __amdgpu_buffer_rsrc_t rimg = make_raw_buffer(img_base, (unsigned)pitch * (unsigned)(pd.H + 2 * kImagePad) * (unsigned)sizeof(IT));
      const int voff = ((iv + (kImagePad - 1)) * pitch + (iu + (kImagePad - 1))) * (int)sizeof(IT);
      bicubic<T>(fu, fv, [&](int l) { return RowLoad<T, IT>::buf(rimg, voff, l * pitch * (int)sizeof(IT)); }, f, Fu, Fv);
    } else {
      const GPtr<IT> base = (GPtr<IT>)(static_cast<const IT *>(img_base) + (size_t)kImagePad * (size_t)pitch + kImagePad) +
                            ((ptrdiff_t)(iv - 1) * pitch + (iu - 1));
      bicubic<T>(fu, fv, [&](int l) { return RowLoad<T, IT>::flat(base + (ptrdiff_t)l * pitch); }, f, Fu, Fv);
    }
  };
  if constexpr (VAR) {
    ProjV<T> pv;
    project_point_var<T>(pd, ps, X, Y, Z, pv);
    state = pv.state;
    sample(pv.iu, pv.iv, pv.fu, pv.fv);
    jacobian_row_var<T>(pd, ps, pv, Fu, Fv, J);
  } else {
    Proj<T> pr;
    project_point<T>(pd, ps, X, Y, Z, pr);
    state = pr.state;
    sample(pr.iu, pr.iv, pr.fu, pr.fv);
    jacobian_row<T>(pd, ps, pr, X, Y, Z, Fu, Fv, J);
  }
  T sc = T(1);
  if (corrected) {  // (uniform)
    T rho, w;
    loss_eval<T>(pd.loss_kind, Uni<T>::loss_a(pd), Uni<T>::loss_inv_b(pd), f * f, rho, w);
    sc = t_sqrt<T>(w);
  }
  const bool bad = state == 2;
  if (__any(bad)) {  // (uniform test first: the common wavefront has nothing to mark)
    if (bad) {
      sc = (T)__builtin_nan("");
      f = T(1);
#pragma unroll
      for (int a = 0; a < 6; ++a) J[a] = T(1);
      if (inb) atomicAdd(n_invalid, 1u);
    }
  }
  f *= sc;
#pragma unroll
  for (int a = 0; a < 6; ++a) J[a] *= sc;
  const long long row0 = pd.row_begin + start;  // first row of this workgroup
  auto put = [&](T *p, T v) { if (nontemporal) __builtin_nontemporal_store(v, p); else *p = v; };
  if (inb) put(r_out + row0 + tid, f);
  if constexpr (LAYOUT == 1) {
    if (inb) {
#pragma unroll
      for (int a = 0; a < 6; ++a) put(J_out + (long long)a * total_rows + row0 + tid, J[a]);
    }
  } else if constexpr (!STAGED) {
    if (inb) {
      Pair<T> *dst = reinterpret_cast<Pair<T> *>(J_out + (row0 + tid) * 6);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Pair<T> v;
        v.x = J[2 * k]; v.y = J[2 * k + 1];
        if (nontemporal) __builtin_nontemporal_store(v, dst + k); else dst[k] = v;
      }
    }
  } else {
    // wavefront-local transpose: lane l writes its 6 values at [6 l .. 6 l + 5] of the wavefront's 384-element strip, then
    // reads elements (2 l, 2 l + 1) + 128 k, k = 0..2, and stores them: each store instruction of the wavefront covers
    // 64 x 2 consecutive elements.  LDS operations of one wavefront complete in order; nothing else touches the strip.
    __shared__ __align__(16) T s_rows[(NT / 64) * 384];
    const int lane = tid & 63, wave = tid >> 6;
    T *strip = s_rows + wave * 384;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Pair<T> v;
      v.x = J[2 * k]; v.y = J[2 * k + 1];
      *reinterpret_cast<Pair<T> *>(strip + 6 * lane + 2 * k) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int valid = 6 * max(0, min(64, count - 64 * wave));  // elements of the strip that belong to real points
    T *dst = J_out + (row0 + 64 * wave) * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = 2 * lane + 128 * k;
      const Pair<T> v = *reinterpret_cast<const Pair<T> *>(strip + e);
      if (e < valid) {
        if (nontemporal) __builtin_nontemporal_store(v, reinterpret_cast<Pair<T> *>(dst + e)); else *reinterpret_cast<Pair<T> *>(dst + e) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The reference's own quantitative self-check (standalone_edge_align.cpp:2494-2567 before the solve, :2704-2776 after):
// every point warped by the pose, divided by its depth, pushed through K, truncated `(int)` to a pixel, the distance
// transform read at that pixel; total, maximum and where the maximum sits.  One partial per workgroup
// {sum, max, u of max, v of max, smallest point index holding the max, points inside, points outside}; the host folds
// them in workgroup order.  Arithmetic is fp64 with the reference's operation order (x / z first, then fx * x + cx, no
// contraction) so that borderline pixels truncate the same way.  Upstream reads outside the image without a check
// (undefined behaviour); such points are skipped here and counted.
struct PixelCostPartial {
  double sum, max, max_u, max_v;
  long long max_index, inside, outside;
  long long pad_;
};

template <typename T>
__global__ __launch_bounds__(kBlockThreads) void ea_pixel_cost_kernel(const ProblemDesc *__restrict__ probs, int problem,
                                                                      const PoseState *__restrict__ poses,
                                                                      PixelCostPartial *__restrict__ out) {
  __shared__ double s_sum[kBlockThreads / 64], s_max[kBlockThreads / 64], s_u[kBlockThreads / 64], s_v[kBlockThreads / 64];
  __shared__ long long s_idx[kBlockThreads / 64], s_in[kBlockThreads / 64], s_out[kBlockThreads / 64];
  const ProblemDesc &pd = probs[problem];
  const PoseState &ps = poses[pd.group];
  const int i = blockIdx.x * kBlockThreads + threadIdx.x;
  double cost = 0.0, mx = -1.0, mu = 0.0, mv = 0.0;  // upstream starts MaxCost at -1
  long long midx = 0x7fffffffffffffffLL, nin = 0, nout = 0;
  if (i < pd.n) {
    const double x = (double)static_cast<const T *>(pd.x)[i], y = (double)static_cast<const T *>(pd.y)[i],
                 z = (double)static_cast<const T *>(pd.z)[i];
    const double *R = ps.R, *t = ps.t;
    // Eigen's 4x4 * 4x1 product, row by row, left to right; then x / z, fx * x + cx.  No contraction: upstream's build
    // has no FMA, and fma(fy, y / z, cy) lands on 34.999999999999986 where the two roundings give 35.0 -- another pixel.
    // (hipcc contracts __dmul_rn + __dadd_rn pairs like plain operators, hence the pragma.)
    double bx, by, bz, u, v;
    {
#pragma clang fp contract(off)
      bx = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
      by = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
      bz = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
      const double xn = bx / bz, yn = by / bz;
      u = pd.fx * xn + pd.cx;
      v = pd.fy * yn + pd.cy;
    }
    const bool finite = (u == u) && (v == v) && fabs(u) < 1e9 && fabs(v) < 1e9;
    const int iu = finite ? (int)u : -1, iv = finite ? (int)v : -1;  // `(int)`: truncation toward zero
    if (finite && iu >= 0 && iu < pd.W && iv >= 0 && iv < pd.H && !(u <= -1.0) && !(v <= -1.0)) {
      cost = (double)static_cast<const T *>(pd.dt)[(size_t)(iv + kImagePad) * (size_t)pd.pitch + (size_t)(iu + kImagePad)];
      mx = cost; mu = u; mv = v; midx = i; nin = 1;
    } else {
      nout = 1;
    }
  }
  // wavefront reduction: sum, count, and (max, smallest index) with the pixel carried along
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    cost += __shfl_xor(cost, m, 64);
    nin += __shfl_xor(nin, m, 64);
    nout += __shfl_xor(nout, m, 64);
    const double omx = __shfl_xor(mx, m, 64), ou = __shfl_xor(mu, m, 64), ov = __shfl_xor(mv, m, 64);
    const long long oi = __shfl_xor(midx, m, 64);
    if (omx > mx || (omx == mx && oi < midx)) { mx = omx; mu = ou; mv = ov; midx = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_sum[wave] = cost; s_max[wave] = mx; s_u[wave] = mu; s_v[wave] = mv; s_idx[wave] = midx; s_in[wave] = nin; s_out[wave] = nout; }
  __syncthreads();
  if (threadIdx.x == 0) {
    PixelCostPartial o;
    o.sum = 0.0; o.max = -1.0; o.max_u = 0.0; o.max_v = 0.0; o.max_index = 0x7fffffffffffffffLL; o.inside = 0; o.outside = 0; o.pad_ = 0;
    for (int w = 0; w < kBlockThreads / 64; ++w) {
      o.sum += s_sum[w]; o.inside += s_in[w]; o.outside += s_out[w];
      if (s_max[w] > o.max || (s_max[w] == o.max && s_idx[w] < o.max_index)) { o.max = s_max[w]; o.max_u = s_u[w]; o.max_v = s_v[w]; o.max_index = s_idx[w]; }
    }
    out[blockIdx.x] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// fixed-order reduction of a problem's tile partials -> 32 accumulators

constexpr int kFoldThreads = 1024;  // plain fold: 32 slots x 32 strided groups
constexpr int kLmThreads = 256;     // fold + scalar LM code: 4 waves = one per SIMD, so the scalar code can keep
                                    // its ~300 live registers without spilling to scratch

// LDS doubles reduce_tiles needs for a workgroup of NTHREADS threads
template <int NTHREADS> constexpr int reduce_tiles_lds() { return (NTHREADS / 16 + NTHREADS / 256) * kAccSlots; }

// A row is 32 doubles = 256 bytes.  Lanes fetch 16 bytes each (two slots): 16 lanes cover a row, a wavefront four rows per
// load instruction.  (With 8 bytes per lane the fold of 196 rows was 128 wave-level load instructions through the one
// texture-address unit of the CU at ~20 cycles each -- the instruction count, not the latency of the round trip, was what
// the LM step waited for: in-kernel stamps, 4.7 k cycles for the fold.)  Thread (id, j) = (tid & 15, tid >> 4) sums slots
// 2 id, 2 id + 1 of rows begin + j + u G, u = 0 .. U-1, G = NTHREADS / 16 row groups, round after round of U G rows.
template <int NTHREADS, int U>
__device__ __forceinline__ void reduce_tiles(const double *__restrict__ partials, int tile_begin,
                                             int tile_end, double *s_part /* reduce_tiles_lds<NTHREADS>() */,
                                             double *out /* 32, threads 0..31 write */) {
  constexpr int G = NTHREADS / 16, F = NTHREADS / 256;
  static_assert(U == 4 || U == 8, "pairwise combine below");
  static_assert(NTHREADS == 256 || NTHREADS == 1024, "final stage: 16 groups per thread");
  typedef double D2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int id = tid & 15, j = tid >> 4;
  // fixed summation order for a given tile count; 2 U independent loads in flight per thread
  D2 s[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = D2{0.0, 0.0};
  // A round = U rows per lane.  Rows past the end are loaded from the last row (a valid address) and masked out
  // of the sum, so a round is U back-to-back loads with no branches; the first two rounds are issued together:
  // folds of up to 2 U G rows cost one memory round trip.
  const int last = tile_end - 1;
  if (tile_begin <= last) {  // uniform
    auto load_round = [&](int base, D2(&v)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = *reinterpret_cast<const D2 *>(partials + (size_t)min(base + j + u * G, last) * kAccSlots + 2 * id);
    };
    auto add_round = [&](int base, const D2(&v)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = base + j + u * G <= last;
        s[u].x += in ? v[u].x : 0.0;
        s[u].y += in ? v[u].y : 0.0;
      }
    };
    D2 v0[U], v1[U];
    load_round(tile_begin, v0);
    if (tile_begin + U * G <= last) {  // uniform
      load_round(tile_begin + U * G, v1);
      add_round(tile_begin, v0);
      add_round(tile_begin + U * G, v1);
      for (int base = tile_begin + 2 * U * G; base <= last; base += U * G) {
        load_round(base, v0);
        add_round(base, v0);
      }
    } else {
      add_round(tile_begin, v0);
    }
  }
#pragma unroll
  for (int w = 1; w < U; w *= 2)
#pragma unroll
    for (int u = 0; u + w < U; u += 2 * w) s[u] += s[u + w];
  *reinterpret_cast<D2 *>(s_part + j * kAccSlots + 2 * id) = s[0];
  __syncthreads();
  // final stage: slot `tid & 31` over 16 row groups per thread, in group order; with 1024 threads four such partial sums per
  // slot, combined in order by the first 32 threads
  if constexpr (F == 1) {
    if (tid < kAccSlots) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < G; ++k) tot += s_part[k * kAccSlots + tid];
      out[tid] = tot;
    }
  } else {
    double *s_fin = s_part + G * kAccSlots;
    if (tid < kAccSlots * F) {
      const int slot = tid & 31, f = tid >> 5;
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += s_part[(16 * f + k) * kAccSlots + slot];
      s_fin[f * kAccSlots + slot] = tot;
    }
    __syncthreads();
    if (tid < kAccSlots) {
      double tot = 0.0;
#pragma unroll
      for (int f = 0; f < F; ++f) tot += s_fin[f * kAccSlots + tid];
      out[tid] = tot;
    }
  }
}

__global__ __launch_bounds__(kFoldThreads) void ea_reduce_kernel(const GroupDesc *__restrict__ groups,
                                                                  const double *__restrict__ partials,
                                                                  EvalOut *__restrict__ out) {
  __shared__ __align__(16) double s_part[reduce_tiles_lds<kFoldThreads>()];
  const GroupDesc gd = groups[blockIdx.x];
  reduce_tiles<kFoldThreads, 4>(partials, gd.tile_begin, gd.tile_end, s_part, out[blockIdx.x].acc);
}

// The same fold for the synchronous evaluations (ea_batch_eval, ea_batch_eval_poses), whose results go straight into pinned
// host memory: the workgroup that finishes last raises a flag there, so that the host can poll for "every result has
// landed" instead of waiting for the stream's completion signal (~10 us later).  Every workgroup's 32 result words are
// stored by its wavefront 0; lane 0 of that wavefront makes them visible system-wide (the fence waits for the wavefront's
// stores), then counts itself in; the workgroup that completes the count re-arms the counter and raises the flag to
// `seq` (release, system scope).  Launches on one stream execute in order: earlier folds of the same call are complete.
__global__ __launch_bounds__(kFoldThreads) void ea_reduce_done_kernel(const GroupDesc *__restrict__ groups,
                                                                       const double *__restrict__ partials,
                                                                       EvalOut *__restrict__ out, unsigned int *__restrict__ counter,
                                                                       int *__restrict__ host_flag, int seq) {
  __shared__ __align__(16) double s_part[reduce_tiles_lds<kFoldThreads>()];
  const GroupDesc gd = groups[blockIdx.x];
  reduce_tiles<kFoldThreads, 4>(partials, gd.tile_begin, gd.tile_end, s_part, out[blockIdx.x].acc);
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The fold as a 256-thread workgroup sums it when it rides in an evaluation launch (ea_eval_fold_kernel below; 8 rows in
// flight per lane keep the riding workgroup inside the evaluation's register budget): closes a pipelined sequence.
__global__ __launch_bounds__(kLmThreads) void ea_reduce256_kernel(const GroupDesc *__restrict__ groups,
                                                                  const double *__restrict__ partials,
                                                                  EvalOut *__restrict__ out) {
  __shared__ __align__(16) double s_part[reduce_tiles_lds<kLmThreads>()];
  const GroupDesc gd = groups[blockIdx.x];
  reduce_tiles<kLmThreads, 4>(partials, gd.tile_begin, gd.tile_end, s_part, out[blockIdx.x].acc);
}

// Evaluation k with the fold of evaluation k-1 riding in the same launch: one extra workgroup per problem (the last
// column of the grid) folds the rows the PREVIOUS launch left in `prev_rows` into `prev_out` while the others evaluate
// into `partials` (a different row array).  A stream of independent evaluations then costs one launch per evaluation
// instead of two dependent ones -- the fold's kernel boundary (1.45 us) and its memory round trip disappear under the
// evaluation.  The fold workgroup sums in the order reduce_tiles<NT> gives a workgroup of this size.
// Plain single-family problems, stencil rows from L2 (MODE 0).
template <typename T, int PPT, int NT, bool BUF, bool IMG32 = false>
__global__ __launch_bounds__(NT) void ea_eval_fold_kernel(
    const void *__restrict__ x0, const void *__restrict__ y0, const void *__restrict__ z0, int n0,
    int shape, int chunks_per_xcd,
    const ProblemDesc *__restrict__ probs, const PoseState *__restrict__ poses,
    double *__restrict__ partials, int lds_texels,
    const GroupDesc *__restrict__ groups, const double *__restrict__ prev_rows, EvalOut *__restrict__ prev_out) {
  if (blockIdx.x == gridDim.x - 1) {  // (uniform)
    __shared__ __align__(16) double s_part[reduce_tiles_lds<NT>()];
    const GroupDesc gd = groups[blockIdx.y];
    reduce_tiles<NT, 4>(prev_rows, gd.tile_begin, gd.tile_end, s_part, prev_out[blockIdx.y].acc);
    return;
  }
  eval_fused_body<T, PPT, 0, NT, false, BUF, IMG32>(x0, y0, z0, n0, shape, chunks_per_xcd, probs, poses, partials, lds_texels);
}

// LM step: fold this problem's partial rows, advance the trust-region state machine, publish the
// next pose to evaluate.  One workgroup per problem.  The state machine is scalar fp64 work on
// lane 0 (pure latency: ~1/3 of an LM iteration), so everything around it is arranged to overlap:
// the state words travel while the partial rows are fetched (sixteen 16-byte loads in flight per lane), the
// host's progress counter is posted before the arithmetic, lane 0 works on a register copy of the
// state, and the pose's float mirrors / the write-back are lane-parallel.
template <int STRAT>
__global__ __launch_bounds__(kLmThreads) void ea_lm_step_kernel(
    const GroupDesc *__restrict__ groups, const double *__restrict__ partials,
    PoseState *__restrict__ poses, LMState *__restrict__ states, LMCold *__restrict__ cold,
    LMTrace *__restrict__ traces, LMOptions opt_arg,
    int *__restrict__ progress /* pinned host: [running x n | evals x n] */,
    LMState *__restrict__ host_states, LMTrace *__restrict__ host_traces /* pinned host, nullable: final delivery */,
    GroupDesc first /* = groups[0], by value: problem 0's row range needs no dependent load */,
    int post_done /* also post "this step is complete" (progress[2 n + p]): the point-sharded solve's look-ahead rule */) {
  __shared__ __align__(16) double s_part[reduce_tiles_lds<kLmThreads>()];
  __shared__ double s_acc[kAccSlots];
  __shared__ LMState s_st;
  __shared__ PoseState s_ps;
  __shared__ int s_trace_it;
  constexpr int kStateWords = (int)(sizeof(LMState) / 8);
  constexpr int kPoseDoubles = 4 + 3 + 9 + 27, kPoseFloats = 9 + 3 + 27;
  static_assert(sizeof(LMState) % 8 == 0 && kStateWords <= kLmThreads, "one 8-byte state word per lane");
  static_assert(offsetof(PoseState, Rf) == kPoseDoubles * 8 && offsetof(PoseState, unit_q) == kPoseDoubles * 8 + kPoseFloats * 4,
                "PoseState layout");
  const int p = blockIdx.x, tid = threadIdx.x;
#ifdef EA_STAMPS
  const int ev_ = states[p].num_evals;
#endif
  EA_LM_STAMP(0, ev_);
#ifdef EA_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0) g_lm_probe_row = ev_ & 63;
#endif
  // Everything the kernel argument segment holds is fetched HERE, in one batch of scalar loads behind one wait: left to
  // the compiler, the options were loaded where lane 0 first uses them -- two more cold misses of the argument segment
  // on the state machine's critical path, after the fold.
  LMOptions opt = opt_arg;
  asm volatile("" : "+s"(opt.max_num_iterations), "+s"(opt.function_tolerance), "+s"(opt.gradient_tolerance),
                    "+s"(opt.parameter_tolerance), "+s"(opt.initial_trust_region_radius), "+s"(opt.max_trust_region_radius),
                    "+s"(opt.min_trust_region_radius), "+s"(opt.min_relative_decrease), "+s"(opt.min_lm_diagonal),
                    "+s"(opt.max_lm_diagonal), "+s"(opt.max_num_consecutive_invalid_steps), "+s"(opt.jacobi_scaling),
                    "+s"(opt.strategy));
  GroupDesc gd = first;
  if (p != 0) gd = groups[p];  // (uniform)
  const int running = states[p].running;
  const int evals_before = states[p].num_evals;  // (a register copy: the LDS copy is rewritten by lane 0 below)
  const double state_word = tid < kStateWords ? reinterpret_cast<const double *>(states + p)[tid] : 0.0;
  EA_LM_STAMP(1, ev_);
  // The rows are fetched WITHOUT waiting for the running flag: the flag's round trip (0.5 us of the 2.5 us this kernel spends
  // before its first arithmetic, in-kernel stamps) then travels beside the rows' instead of in front of it.  A finished
  // problem folds rows nobody reads and leaves below.
  reduce_tiles<kLmThreads, 8>(partials, gd.tile_begin, gd.tile_end, s_part, s_acc);
  if (!running) return;  // uniform
  if (tid < kStateWords) reinterpret_cast<double *>(&s_st)[tid] = state_word;
  __syncthreads();
  EA_LM_STAMP(2, ev_);
  // the host only uses this counter to decide how far ahead to enqueue: posted by another wavefront before the
  // arithmetic, so the PCIe write is neither the last thing the kernel waits for nor in lane 0's memory counter
  if (tid == 64)
    __hip_atomic_store(progress + gridDim.x + p, evals_before + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  LMPending pend;
  double acc[kAccSlots];
  if (tid == 0) {
    LMState st;
    lm_copy_state(&st, &s_st);
#pragma unroll
    for (int i = 0; i < kAccSlots; ++i) acc[i] = s_acc[i];
    if (EA_UNLIKELY(st.num_evals == 0)) lm_begin<STRAT>(&st, cold + p, traces + p, &opt, acc, &pend);
    else lm_advance<STRAT>(&st, cold + p, traces + p, &opt, acc, &pend);
    EA_LM_STAMP(3, ev_);
    make_pose_core(st.cand, st.rot_transposed, st.running, &s_ps, /*zero_unused_G=*/false);
    lm_copy_state(&s_st, &st);
    s_trace_it = pend.trace_it;
    EA_LM_STAMP(4, ev_);
  }
  __syncthreads();
  if (tid < kStateWords) reinterpret_cast<double *>(states + p)[tid] = reinterpret_cast<const double *>(&s_st)[tid];
  {
    // pose: doubles copied by lanes 64.., float mirrors converted by lanes 128.., flags by lane 192
    // (G is never read when unit_q is set; those lanes publish zeros)
    const double *pd_src = reinterpret_cast<const double *>(&s_ps);
    const int a = tid - 64, f = tid - 128;
    const bool skip_g = s_ps.unit_q != 0;
    if (a >= 0 && a < kPoseDoubles) reinterpret_cast<double *>(poses + p)[a] = (skip_g && a >= 16) ? 0.0 : pd_src[a];
    if (f >= 0 && f < kPoseFloats) {
      // Rf | tf | Gf mirror R | t | G
      const double v = f < 9 ? s_ps.R[f] : (f < 12 ? s_ps.t[f - 9] : (skip_g ? 0.0 : s_ps.G[f - 12]));
      (&poses[p].Rf[0])[f] = (float)v;
    }
    if (tid == 192) { poses[p].unit_q = s_ps.unit_q; poses[p].active = s_ps.active; }
  }
  if (tid == 0) lm_flush(&pend, cold + p, traces + p, acc);
  if (!s_st.running) {  // uniform: the solve of this problem ends with this launch
    // Deliver the result straight into pinned host memory -- final state and the trace rows written so far -- and
    // only then lower the flag the host polls: the host returns without a device-to-host copy or a stream
    // synchronisation behind launches that were queued ahead and now have nothing to do.
    if (host_states) {
      if (tid < kStateWords) reinterpret_cast<double *>(host_states + p)[tid] = reinterpret_cast<const double *>(&s_st)[tid];
      const int last = min(s_st.iteration, kTrace - 1);  // rows 0 .. last exist
      auto copy_row = [&](int r) {
        const LMTrace &src = traces[p];
        LMTrace &dst = host_traces[p];
        dst.it_cost[r] = src.it_cost[r];
        dst.it_cost_change[r] = src.it_cost_change[r];
        dst.it_gradient_max_norm[r] = src.it_gradient_max_norm[r];
        dst.it_step_norm[r] = src.it_step_norm[r];
        dst.it_relative_decrease[r] = src.it_relative_decrease[r];
        dst.it_radius[r] = src.it_radius[r];
        dst.it_successful[r] = src.it_successful[r];
      };
      if (s_trace_it == last) {
        // the usual end: rows < last come from earlier launches (one lane each), lane 0 holds the last one
        if (tid < last) copy_row(tid);
        if (tid == 0) {
          LMPending h = pend;
          h.store_system = 0;
          lm_flush(&h, cold + p, host_traces + p, acc);
        }
      } else if (tid == 0) {
        // rows written by this very launch through lm_trace (invalid steps): lane 0 wrote them, lane 0 copies them
        __threadfence();
        for (int r = 0; r <= last; ++r) copy_row(r);
      }
      __threadfence_system();
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(progress + p, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // behind the flag (same lane, release order): a host that has seen "step k complete" sees the flag as step k left it
  if (post_done && tid == 0)
    __hip_atomic_store(progress + 2 * gridDim.x + p, evals_before + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef EA_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  EA_LM_STAMP(5, ev_);
}

// ---- one launch per LM iteration ----------------------------------------------------------------------------------------
//
// ea_solve's loop as (evaluate, step) pairs pays two kernel boundaries per iteration and, in between, the publication of the
// candidate pose through memory: evaluation kernel ends -> step kernel starts cold on one CU, folds, runs the state machine,
// stores pose and state, ends -> evaluation kernel starts cold, fetches descriptor and pose, ...  For ONE small problem the
// evaluation is at most one workgroup per CU, and every CU but one idles through the step.  Here every workgroup of the
// evaluation runs the step itself: it folds the previous launch's partial rows (the same reduce_tiles in the same order),
// runs the state machine on its lane 0 -- identical instructions on identical inputs, so every workgroup arrives at the same
// candidate pose without talking to the others -- and evaluates its own points at that pose straight away, its point loads
// having travelled beside the fold.  One kernel boundary per iteration instead of two, no pose round trip through memory, and
// the cold start of the evaluation (descriptor, points) hidden behind the fold.  Nothing is handed from one workgroup to
// another inside a launch; what a launch reads (state, cold system, rows) it never writes: those three are ping-ponged
// between two buffers by the parity of the launch.
//
// One more workgroup than chunks: the one past the last chunk has no points and is the WRITER -- state, cold system, trace row,
// the pose (for whoever evaluates at it afterwards), the host's progress words and, when the solve ends, the final delivery
// into pinned host memory, exactly as ea_lm_step_kernel does them.  Its stores do not sit in any evaluator's memory counter.
// A finished problem is recognised by every later launch from poses[p].active == 0 (written once, by the writer of the
// launch in which the state machine stopped; in that launch every workgroup finds out by itself).
// The cold system (JtJ, Jtr at x for the step after a rejected one; the dogleg's vectors) is read into LDS, one copy per
// wavefront, and the state machine works on that copy; the writer stores its copy for the next launch.
// One plain residual family per problem, 256-thread workgroups, stencil rows from L2; the host only takes this path when the
// whole grid is resident at once (<= 256 workgroups), see solve_start.
template <typename T, int PPT, bool BUF, bool IMG32, int STRAT>
__global__ __launch_bounds__(kLmThreads) void ea_lm_iter_kernel(
    // the 14 preloaded dwords: what problem 0's point loads and ROW loads need -- the fold is the head of every workgroup's
    // dependent chain, and its loads must not wait for a scalar load of the argument segment (tile0_* = problem 0's row range)
    const void *__restrict__ x0, const void *__restrict__ y0, const void *__restrict__ z0, int n0,
    int shape, int chunks_per_xcd, const double *__restrict__ rows_in, int tile0_begin, int tile0_end,
    const ProblemDesc *__restrict__ probs, PoseState *__restrict__ poses,
    double *__restrict__ rows_out, const GroupDesc *__restrict__ groups,
    const LMState *__restrict__ st_in, LMState *__restrict__ st_out, const LMCold *__restrict__ cold_in,
    LMCold *__restrict__ cold_out, LMTrace *__restrict__ traces, LMOptions opt_arg,
    int *__restrict__ progress /* pinned host: [running x n | evals x n | steps complete x n] */,
    LMState *__restrict__ host_states, LMTrace *__restrict__ host_traces,
    int post_done /* the writer also posts "this step is complete" (progress[2 n + p]): the point-sharded solve's look-ahead rule */) {
  constexpr int NT = kLmThreads;
  __shared__ __align__(16) double s_part[reduce_tiles_lds<NT>()];
  __shared__ double s_acc[kAccSlots];
  __shared__ LMState s_st;
  __shared__ PoseState s_pose[NT / 64];  // one per wavefront (the writer uses the first)
  __shared__ LMCold s_cold[NT / 64];
  __shared__ int s_trace_it, s_store_system;
  extern __shared__ __align__(16) unsigned char smem[];
  double *s_red = reinterpret_cast<double *>(smem);
  int *s_box = reinterpret_cast<int *>(smem + kRedBytes);
  constexpr int kStateWords = (int)(sizeof(LMState) / 8), kColdWords = (int)(sizeof(LMCold) / 8);
  constexpr int kPoseDoubles = 4 + 3 + 9 + 27, kPoseFloats = 9 + 3 + 27;
  static_assert(kStateWords <= NT && kColdWords <= NT, "one 8-byte word per lane");
  const int chunk = shape & 0xffff, xcd_remap = (shape >> 16) & 1;
  const int p = blockIdx.y, tid = threadIdx.x;
  const int bx = blockIdx.x;
  const int c = xcd_remap ? (bx & 7) * chunks_per_xcd + (bx >> 3) : bx;
  const bool writer = bx == (int)gridDim.x - 1;  // (= the last chunk index in either mapping; the launcher sizes the grid past the data)
  const long long start = (long long)c * chunk;
  LMOptions opt = opt_arg;
  asm volatile("" : "+s"(opt.max_num_iterations), "+s"(opt.function_tolerance), "+s"(opt.gradient_tolerance),
                    "+s"(opt.parameter_tolerance), "+s"(opt.initial_trust_region_radius), "+s"(opt.max_trust_region_radius),
                    "+s"(opt.min_trust_region_radius), "+s"(opt.min_relative_decrease), "+s"(opt.min_lm_diagonal),
                    "+s"(opt.max_lm_diagonal), "+s"(opt.max_num_consecutive_invalid_steps), "+s"(opt.jacobi_scaling),
                    "+s"(opt.strategy));
  // problem 0's points go out first (their addresses came with the wave), then everything uniform in one batch
  T X[PPT], Y[PPT], Z[PPT];
  const bool early = BUF && p == 0 && n0 > 0 && start < n0;  // (uniform)
  if constexpr (BUF) {
    if (early) {
      const int count0 = min(chunk, (int)(n0 - start));
      const __amdgpu_buffer_rsrc_t rx = make_raw_buffer(static_cast<const T *>(x0) + start, (unsigned)count0 * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t ry = make_raw_buffer(static_cast<const T *>(y0) + start, (unsigned)count0 * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t rz = make_raw_buffer(static_cast<const T *>(z0) + start, (unsigned)count0 * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int poff = min(tid + k * NT, count0 - 1) * (int)sizeof(T);
        X[k] = buf_load_elem<T>(rx, poff); Y[k] = buf_load_elem<T>(ry, poff); Z[k] = buf_load_elem<T>(rz, poff);
      }
    }
  }
  EA_LM_CLOCK(t_enter_);  // (diagnostic build: kernel entered)
  const ProblemDesc pd = probs[p];
  const int active = poses[p].active;
  GroupDesc gd = {tile0_begin, tile0_end, 0, 1};
  if (p != 0) gd = groups[p];  // (uniform)
  const int evals_before = st_in[p].num_evals;
  const double state_word = tid < kStateWords ? reinterpret_cast<const double *>(st_in + p)[tid] : 0.0;
  const double cold_word = (tid & 63) < kColdWords ? reinterpret_cast<const double *>(cold_in + p)[tid & 63] : 0.0;
  // ---- the step, in every workgroup: fold of the previous launch's rows + the state machine on lane 0.
  // The rows are fetched at once, beside the uniforms above and not behind them (as ea_lm_step_kernel does): a launch that
  // finds its problem finished has folded rows nobody reads and leaves below.
  reduce_tiles<NT, 8>(rows_in, gd.tile_begin, gd.tile_end, s_part, s_acc);
  asm volatile("" ::"s"(pd.x), "s"(pd.y), "s"(pd.z), "s"(IMG32 ? pd.dt32 : pd.dt), "s"(pd.n), "s"(pd.W), "s"(pd.H), "s"(pd.pitch),
               "s"(Uni<T>::fx(pd)), "s"(Uni<T>::fy(pd)), "s"(Uni<T>::cx(pd)), "s"(Uni<T>::cy(pd)),
               "s"(Uni<T>::loss_a(pd)), "s"(Uni<T>::loss_inv_b(pd)), "s"(Uni<T>::z_guard(pd)), "s"(Uni<T>::z_eps(pd)),
               "s"(pd.loss_kind), "s"(pd.tile_begin), "s"(active), "s"(evals_before));
  if (!active) return;  // the solve of this problem ended in an earlier launch
  const bool evaluator = start < pd.n;
  if (!evaluator && !writer) return;
  const int count = evaluator ? min(chunk, (int)(pd.n - start)) : 0;
  if (evaluator && !early) {  // (problems behind the first of a batch, or flat addressing: their points wait for the descriptor)
    const T *px = static_cast<const T *>(pd.x) + start, *py = static_cast<const T *>(pd.y) + start, *pz = static_cast<const T *>(pd.z) + start;
    if constexpr (BUF) {
      const __amdgpu_buffer_rsrc_t rx = make_raw_buffer(px, (unsigned)count * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t ry = make_raw_buffer(py, (unsigned)count * (unsigned)sizeof(T));
      const __amdgpu_buffer_rsrc_t rz = make_raw_buffer(pz, (unsigned)count * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int poff = min(tid + k * NT, count - 1) * (int)sizeof(T);
        X[k] = buf_load_elem<T>(rx, poff); Y[k] = buf_load_elem<T>(ry, poff); Z[k] = buf_load_elem<T>(rz, poff);
      }
    } else {
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int jj = min(tid + k * NT, count - 1);
        X[k] = ((GPtr<T>)px)[jj]; Y[k] = ((GPtr<T>)py)[jj]; Z[k] = ((GPtr<T>)pz)[jj];
      }
    }
  }
  if (tid < kStateWords) reinterpret_cast<double *>(&s_st)[tid] = state_word;
  if ((tid & 63) < kColdWords) reinterpret_cast<double *>(&s_cold[tid >> 6])[tid & 63] = cold_word;  // (own wavefront's copy)
  __syncthreads();
  EA_LM_STAMP_PUT(0, evals_before, t_enter_);
  EA_LM_STAMP(1, evals_before);  // folded
  if (writer) {
    // ---- the writer: the full state machine on lane 0, then everything that is stored
    if (tid == 64)
      __hip_atomic_store(progress + gridDim.y + p, evals_before + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    LMPending pend;
    double acc[kAccSlots];
    if (tid == 0) {
      LMState st;
      lm_copy_state(&st, &s_st);
#pragma unroll
      for (int i = 0; i < kAccSlots; ++i) acc[i] = s_acc[i];
      if (EA_UNLIKELY(st.num_evals == 0)) lm_begin<STRAT>(&st, &s_cold[0], traces + p, &opt, acc, &pend);
      else lm_advance<STRAT>(&st, &s_cold[0], traces + p, &opt, acc, &pend);
      make_pose_core(st.cand, st.rot_transposed, st.running, &s_pose[0], /*zero_unused_G=*/false);
      lm_copy_state(&s_st, &st);
      s_trace_it = pend.trace_it;
      s_store_system = pend.store_system;
    }
    __syncthreads();
    const int running = s_st.running;
    if (tid < kStateWords) reinterpret_cast<double *>(st_out + p)[tid] = reinterpret_cast<const double *>(&s_st)[tid];
    {
      const double *pd_src = reinterpret_cast<const double *>(&s_pose[0]);
      const int a = tid - 64, f = tid - 128;
      const bool skip_g = s_pose[0].unit_q != 0;
      if (a >= 0 && a < kPoseDoubles) reinterpret_cast<double *>(poses + p)[a] = (skip_g && a >= 16) ? 0.0 : pd_src[a];
      if (f >= 0 && f < kPoseFloats) {
        const double v = f < 9 ? s_pose[0].R[f] : (f < 12 ? s_pose[0].t[f - 9] : (skip_g ? 0.0 : s_pose[0].G[f - 12]));
        (&poses[p].Rf[0])[f] = (float)v;
      }
      if (tid == 192) { poses[p].unit_q = s_pose[0].unit_q; poses[p].active = s_pose[0].active; }
    }
    // the cold system of the next launch: JtJ, Jtr of this evaluation (accepted step: lm_flush) or what this launch read
    // (rejected step); the dogleg's vectors as the state machine left them
    if (tid < kColdWords && !(s_store_system && tid < 27))
      reinterpret_cast<double *>(cold_out + p)[tid] = reinterpret_cast<const double *>(&s_cold[0])[tid];
    if (tid == 0) lm_flush(&pend, cold_out + p, traces + p, acc);
    if (!running) {
      if (host_states) {
        if (tid < kStateWords) reinterpret_cast<double *>(host_states + p)[tid] = reinterpret_cast<const double *>(&s_st)[tid];
        const int last = min(s_st.iteration, kTrace - 1);
        auto copy_row = [&](int r) {
          const LMTrace &src = traces[p];
          LMTrace &dst = host_traces[p];
          dst.it_cost[r] = src.it_cost[r];
          dst.it_cost_change[r] = src.it_cost_change[r];
          dst.it_gradient_max_norm[r] = src.it_gradient_max_norm[r];
          dst.it_step_norm[r] = src.it_step_norm[r];
          dst.it_relative_decrease[r] = src.it_relative_decrease[r];
          dst.it_radius[r] = src.it_radius[r];
          dst.it_successful[r] = src.it_successful[r];
        };
        if (s_trace_it == last) {
          if (tid < last) copy_row(tid);
          if (tid == 0) {
            LMPending h = pend;
            h.store_system = 0;
            lm_flush(&h, cold_out + p, host_traces + p, acc);
          }
        } else if (tid == 0) {
          __threadfence();
          for (int r = 0; r <= last; ++r) copy_row(r);
        }
        __threadfence_system();
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(progress + p, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // Point-sharded solve (ea_solve_sharded_comm): the rows of this launch are all-reduced over the ranks in place and the
    // next launch folds the range the widest rank needs; this rank's rows past its own stay zero -- but the all-reduce left
    // sums there two launches ago, so the writer clears them again.
    for (int r = pd.tile_end + (tid >> 5); r < gd.tile_end; r += NT / 32) rows_out[(size_t)r * kAccSlots + (tid & 31)] = 0.0;
    if (post_done && tid == 0)  // behind the flag (same lane, release order), as in ea_lm_step_kernel
      __hip_atomic_store(progress + 2 * gridDim.y + p, evals_before + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // ---- the evaluators: every WAVEFRONT takes the step itself, in the form that yields the candidate pose and nothing else
  // (lm_take_system<LITE>) -- four identical computations on four SIMDs, so the pose reaches the lanes of a wavefront without a
  // workgroup barrier: lane 0 leaves it in the wavefront's own LDS slot and its wavefront reads it back (LDS operations of one
  // wavefront execute in order).
  PoseState &s_ps = s_pose[tid >> 6];
  int running_v = 0;
  if ((tid & 63) == 0) {
    LMState st;
    LMPending pend;
    double acc[kAccSlots];
    lm_copy_state(&st, &s_st);
#pragma unroll
    for (int i = 0; i < kAccSlots; ++i) acc[i] = s_acc[i];
    LMCold *cold = &s_cold[tid >> 6];
    if (EA_UNLIKELY(st.num_evals == 0)) lm_begin<STRAT, true>(&st, cold, nullptr, &opt, acc, &pend);
    else lm_advance<STRAT, true>(&st, cold, nullptr, &opt, acc, &pend);
    EA_LM_STAMP(2, evals_before);  // state machine done
    make_pose_core(st.cand, st.rot_transposed, st.running, &s_ps, /*zero_unused_G=*/false);
    running_v = st.running;
  }
  const int running = __builtin_amdgcn_readfirstlane(running_v);
  EA_LM_STAMP(3, evals_before);  // pose in LDS
  if (!running) return;  // (uniform: every workgroup of the problem arrived at the same state)
  // ---- the evaluation at the candidate pose
  if (!s_ps.unit_q) {  // (uniform, rare) the general-quaternion Jacobian reads G through the pose pointer
    if ((tid & 63) < 27) s_ps.Gf[tid & 63] = (float)s_ps.G[tid & 63];
  }
  PoseLite<T> ps;
#pragma unroll
  for (int i = 0; i < 9; ++i) ps.R[i] = uniform_scalar<T>((T)s_ps.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) ps.t[i] = uniform_scalar<T>((T)s_ps.t[i]);
  ps.unit_q = __builtin_amdgcn_readfirstlane(s_ps.unit_q);
  ps.full = &s_ps;
  const double sum = fused_chunk<T, PPT, 0, NT, false, BUF, IMG32, PoseLite<T>>(pd, ps, X, Y, Z, count, s_red, s_box, (T *)nullptr, 0,
                                                                            tid < kAccSlots ? tid : -1);
  EA_LM_STAMP(4, evals_before);  // evaluated
  if (tid < kAccSlots) rows_out[(size_t)(pd.tile_begin + c) * kAccSlots + tid] = sum;
#ifdef EA_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  EA_LM_STAMP(5, evals_before);  // row stored
}

// q (w, x, y, z) and t of n = K x count poses, 7 doubles each -> the PoseState the evaluation kernels read, built on the
// device (ea_batch_set_poses: K different poses per problem go up as 56 bytes each instead of a 600-byte PoseState the
// host would have to compute).  Pose i belongs to problem i % count; its rotation is applied transposed for the ROS
// flavour (the problem's first term says so).
__global__ __launch_bounds__(64) void ea_make_poses_kernel(const double *__restrict__ qt, int n, int count,
                                                           const ProblemDesc *__restrict__ probs,
                                                           const GroupDesc *__restrict__ groups, PoseState *__restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double x[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) x[k] = qt[(size_t)i * 7 + k];
  PoseState *ps = out + i;
  make_pose_state(x, probs[groups[i % count].term_begin].rot_transposed, 1, ps);
  ps->pad_[0] = 0; ps->pad_[1] = 0;
}

// pad + convert a row-major [H][W] device image into the replicated-border layout
// (dst32, inexact: fp64 problems only, nullable -- the float32 mirror of the image and a flag raised by any value the
// mirror does not hold exactly; ProblemDesc::dt32)
template <typename T>
__global__ void ea_pad_image_kernel(const T *__restrict__ src, int H, int W, T *__restrict__ dst, int pitch,
                                    float *__restrict__ dst32, int *__restrict__ inexact) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;  // padded coords
  const int v = blockIdx.y;
  if (u >= W + 2 * kImagePad) return;
  const int su = min(max(u - kImagePad, 0), W - 1);
  const int sv = min(max(v - kImagePad, 0), H - 1);
  const T val = src[(size_t)sv * W + su];
  dst[(size_t)v * pitch + u] = val;
  if (dst32) {
    const float f = (float)val;
    dst32[(size_t)v * pitch + u] = f;
    if (!((T)f == val)) atomicOr(inexact, 1);  // (also NaN)
  }
}

// The reference's Grid2D view (rows index u, columns index v: data[u * H + v], doubles; standalone_edge_align.cpp:258) ->
// the padded image in the problem's dtype: transpose, replicate the border, convert.  A 32 x 32 tile goes through LDS so
// that both the reads (v contiguous in the source) and the writes (u contiguous in the image) are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void ea_grid_to_image_kernel(const double *__restrict__ grid, int W, int H, T *__restrict__ dst, int pitch,
                                                               float *__restrict__ dst32, int *__restrict__ inexact) {
  __shared__ double s_tile[32][33];
  const int PW = W + 2 * kImagePad, PH = H + 2 * kImagePad;
  const int u0 = blockIdx.x * 32, v0 = blockIdx.y * 32;   // padded coordinates of the tile
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // source element (u, v): v runs with tx (contiguous in the grid), u with ty + 8 k
    const int u = u0 + ty + 8 * k, v = v0 + tx;
    const int su = min(max(u - kImagePad, 0), W - 1), sv = min(max(v - kImagePad, 0), H - 1);
    s_tile[ty + 8 * k][tx] = grid[(size_t)su * H + sv];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = v0 + ty + 8 * k, u = u0 + tx;
    if (v < PH && u < PW) {
      const double val = s_tile[tx][ty + 8 * k];
      dst[(size_t)v * pitch + u] = (T)val;
      if (dst32) {
        const float f = (float)val;
        dst32[(size_t)v * pitch + u] = f;
        if (!((double)f == val)) atomicOr(inexact, 1);  // (also NaN)
      }
    }
  }
}

// the reference's AoS points (columns of a_X: x y z [w] per point, doubles; utils.cpp:268-280) -> SoA in the problem's dtype
template <typename T>
__global__ __launch_bounds__(256) void ea_aos_to_soa_kernel(const double *__restrict__ src, long long n, int stride, T *__restrict__ x,
                                                            T *__restrict__ y, T *__restrict__ z) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double *s = src + i * stride;
  x[i] = (T)s[0]; y[i] = (T)s[1]; z[i] = (T)s[2];
}
#endif  // !EA_TU_VARIANT

// ------------------------------------------------------------------------------------------------
// launchers (called from ea_capi.cpp)

// The fused evaluation lives in two translation units: this file as it is (plain functor, every launch shape) and the
// same file compiled with -DEA_TU_VARIANT through ea_kernels_var.hip (the distortion / second-camera functors only).
// They differ in ONE compiler setting: the plain kernels are scheduled for instruction-level parallelism
// (-mllvm -amdgpu-sched-strategy=max-ilp: -2 .. -4 % kernel time), which costs the variant kernels a wave of occupancy in
// fp64 (123 -> 136 VGPRs) and 4-7 % of their time -- they keep the default strategy (build.py; profiles/LOG.md section 5b).
#define EA_LAUNCH_B(T, P, L, N, V, B)                                                              \
  hipLaunchKernelGGL((EvalKernel<TAG, T, P, L, N, V, B, false>::fn), grid, dim3(N), shmem, stream, x0, y0, z0, n0, shape, \
                     chunks_per_xcd, probs, poses, partials, lds_texels)
#define EA_LAUNCH(T, P, L, N, V)                                                                   \
  do {                                                                                             \
    if (buffer_loads && (L) == 0) EA_LAUNCH_B(T, P, 0, N, V, true); else EA_LAUNCH_B(T, P, L, N, V, false); \
  } while (0)
#define EA_LAUNCH_PROLOGUE                                                                          \
  if (nterms <= 0 || max_chunks <= 0) return hipSuccess;                                            \
  const int chunks_per_xcd = (max_chunks + 7) / 8;                                                  \
  const dim3 grid(xcd_remap ? chunks_per_xcd * 8 : max_chunks, nterms);                             \
  if (chunk <= 0 || chunk > 0xffff) return hipErrorInvalidValue; /* (NT * PPT <= 4096) */           \
  const int shape = chunk | ((xcd_remap ? 1 : 0) << 16) | ((terms_are_groups ? 1 : 0) << 17);       \
  const int esz = dtype == 1 ? 4 : 8;                                                               \
  const int lds_texels = lds_bytes > 0 ? lds_bytes / esz : 0;                                       \
  const size_t shmem = (size_t)kHdrBytes + (size_t)lds_texels * esz;

#ifdef EA_TU_VARIANT
// distortion / second-camera terms: 256-thread workgroups, L2 path, 1-2 points per lane
template <int TAG>
static hipError_t launch_eval_fused_var_t(int dtype, int ppt, const ProblemDesc *probs, int nterms, int chunk, int max_chunks,
                                          int xcd_remap, const PoseState *poses, double *partials, int terms_are_groups,
                                          int buffer_loads, const void *x0, const void *y0, const void *z0, int n0,
                                          hipStream_t stream) {
  const int lds_bytes = 0;
  EA_LAUNCH_PROLOGUE
  if (dtype == 1) { if (ppt == 1) EA_LAUNCH(float, 1, 0, 256, true); else EA_LAUNCH(float, 2, 0, 256, true); }
  else { if (ppt == 1) EA_LAUNCH(double, 1, 0, 256, true); else EA_LAUNCH(double, 2, 0, 256, true); }
  return hipGetLastError();
}
// tag 1: the launch of ea_batch_eval_poses (kernel name ea_eval_poses_kernel)
hipError_t launch_eval_fused_var(int tag, int dtype, int ppt, const ProblemDesc *probs, int nterms, int chunk, int max_chunks,
                                 int xcd_remap, const PoseState *poses, double *partials, int terms_are_groups,
                                 int buffer_loads, const void *x0, const void *y0, const void *z0, int n0,
                                 hipStream_t stream) {
  return tag ? launch_eval_fused_var_t<1>(dtype, ppt, probs, nterms, chunk, max_chunks, xcd_remap, poses, partials, terms_are_groups,
                                          buffer_loads, x0, y0, z0, n0, stream)
             : launch_eval_fused_var_t<0>(dtype, ppt, probs, nterms, chunk, max_chunks, xcd_remap, poses, partials, terms_are_groups,
                                          buffer_loads, x0, y0, z0, n0, stream);
}
#else
hipError_t launch_eval_fused_var(int tag, int dtype, int ppt, const ProblemDesc *probs, int nterms, int chunk, int max_chunks,
                                 int xcd_remap, const PoseState *poses, double *partials, int terms_are_groups,
                                 int buffer_loads, const void *x0, const void *y0, const void *z0, int n0,
                                 hipStream_t stream);  // ea_kernels_var.hip

template <int TAG>
static hipError_t launch_eval_fused_t(int dtype, int ppt, int nt, int variant, const ProblemDesc *probs, int nterms, int chunk,
                                      int max_chunks, int xcd_remap, const PoseState *poses, double *partials,
                                      int lds_bytes, int wide, int terms_are_groups, int buffer_loads, int img32, const void *x0, const void *y0,
                                      const void *z0, int n0, hipStream_t stream) {
  if (variant)
    return launch_eval_fused_var(TAG, dtype, ppt, probs, nterms, chunk, max_chunks, xcd_remap, poses, partials, terms_are_groups,
                                 buffer_loads, x0, y0, z0, n0, stream);
  EA_LAUNCH_PROLOGUE
  if (img32) {
    // fp64 arithmetic over the fp32-stored image (every term has ProblemDesc::dt32): the L2 path only
    if (dtype != 0 || lds_texels > 0) return hipErrorInvalidValue;
#define EA_LAUNCH_I(P, N)                                                                                              \
  do {                                                                                                                 \
    if (buffer_loads)                                                                                                  \
      hipLaunchKernelGGL((EvalKernel<TAG, double, P, 0, N, false, true, true>::fn), grid, dim3(N), shmem, stream, x0, y0, z0, n0, shape, \
                         chunks_per_xcd, probs, poses, partials, lds_texels);                                          \
    else                                                                                                               \
      hipLaunchKernelGGL((EvalKernel<TAG, double, P, 0, N, false, false, true>::fn), grid, dim3(N), shmem, stream, x0, y0, z0, n0, shape, \
                         chunks_per_xcd, probs, poses, partials, lds_texels);                                          \
  } while (0)
    if (nt == 1024) EA_LAUNCH_I(1, 1024);
    else if (ppt == 1) EA_LAUNCH_I(1, 256);
    else EA_LAUNCH_I(2, 256);
#undef EA_LAUNCH_I
    return hipGetLastError();
  }
#define EA_LAUNCH_L(T, P, N)                                                          \
  do {                                                                                \
    if (lds_texels > 0) EA_LAUNCH(T, P, 1, N, false); else EA_LAUNCH(T, P, 0, N, false); \
  } while (0)
  if (dtype == 1 && wide && lds_texels == 0) {
    // fp64 sums from the lane's sum on (MODE 2; ea_batch_set_tuning "wide_accumulate")
#define EA_LAUNCH_W(P, N) do { if (buffer_loads) EA_LAUNCH_B(float, P, 2, N, false, true); else EA_LAUNCH_B(float, P, 2, N, false, false); } while (0)
    if (nt == 1024) { if (ppt == 1) EA_LAUNCH_W(1, 1024); else if (ppt == 2) EA_LAUNCH_W(2, 1024); else EA_LAUNCH_W(4, 1024); }
    else if (ppt == 1) EA_LAUNCH_W(1, 256);
    else if (ppt == 2) EA_LAUNCH_W(2, 256);
    else EA_LAUNCH_W(4, 256);
#undef EA_LAUNCH_W
  } else if (dtype == 1) {
    if (nt == 1024) { if (ppt == 1) EA_LAUNCH_L(float, 1, 1024); else if (ppt == 2) EA_LAUNCH_L(float, 2, 1024); else EA_LAUNCH_L(float, 4, 1024); }
    else if (ppt == 1) EA_LAUNCH_L(float, 1, 256);
    else if (ppt == 2) EA_LAUNCH_L(float, 2, 256);
    else EA_LAUNCH_L(float, 4, 256);
  } else {
    if (nt == 1024) EA_LAUNCH_L(double, 1, 1024);
    else if (ppt == 1) EA_LAUNCH_L(double, 1, 256);
    else EA_LAUNCH_L(double, 2, 256);
  }
#undef EA_LAUNCH_L
  return hipGetLastError();
}

hipError_t launch_eval_fused(int dtype, int ppt, int nt, int variant, const ProblemDesc *probs, int nterms, int chunk,
                             int max_chunks, int xcd_remap, const PoseState *poses, double *partials,
                             int lds_bytes, int wide, int terms_are_groups, int buffer_loads, int img32, const void *x0, const void *y0,
                             const void *z0, int n0, hipStream_t stream) {
  return launch_eval_fused_t<0>(dtype, ppt, nt, variant, probs, nterms, chunk, max_chunks, xcd_remap, poses, partials, lds_bytes, wide,
                                terms_are_groups, buffer_loads, img32, x0, y0, z0, n0, stream);
}
// the same launch under the kernel name ea_eval_poses_kernel: G poses x terms in grid y (ea_batch_eval_poses)
hipError_t launch_eval_poses(int dtype, int ppt, int nt, int variant, const ProblemDesc *probs, int nterms, int chunk,
                             int max_chunks, int xcd_remap, const PoseState *poses, double *partials,
                             int lds_bytes, int wide, int terms_are_groups, int buffer_loads, int img32, const void *x0, const void *y0,
                             const void *z0, int n0, hipStream_t stream) {
  return launch_eval_fused_t<1>(dtype, ppt, nt, variant, probs, nterms, chunk, max_chunks, xcd_remap, poses, partials, lds_bytes, wide,
                                terms_are_groups, buffer_loads, img32, x0, y0, z0, n0, stream);
}

hipError_t launch_reduce_nt(int nt, const GroupDesc *groups, int count, const double *partials, EvalOut *out,
                            hipStream_t stream);

// evaluation into `partials` + the fold of `prev_rows` -> `prev_out` in one launch (ea_eval_fold_kernel)
hipError_t launch_eval_fold(int dtype, int ppt, int nt, const ProblemDesc *probs, int nterms, int chunk, int max_chunks,
                            int xcd_remap, const PoseState *poses, double *partials, int buffer_loads, int img32, const void *x0,
                            const void *y0, const void *z0, int n0, const GroupDesc *groups, const double *prev_rows,
                            EvalOut *prev_out, hipStream_t stream) {
  const int lds_bytes = 0, terms_are_groups = 1;
  // a batch without a single point has nothing to evaluate, but the previous step's fold is still owed (its result slot
  // must not keep what an earlier owner of the memory left there): the stand-alone fold in the riders' summation order
  if (nterms > 0 && max_chunks <= 0) return launch_reduce_nt(nt, groups, nterms, prev_rows, prev_out, stream);
  EA_LAUNCH_PROLOGUE
  const dim3 grid_f(grid.x + 1, grid.y);
  if (img32) {
    if (dtype != 0) return hipErrorInvalidValue;
#define EA_LAUNCH_FI(P, N)                                                                                            \
  do {                                                                                                                \
    if (buffer_loads)                                                                                                 \
      hipLaunchKernelGGL((ea_eval_fold_kernel<double, P, N, true, true>), grid_f, dim3(N), shmem, stream, x0, y0, z0, n0, shape,  \
                         chunks_per_xcd, probs, poses, partials, lds_texels, groups, prev_rows, prev_out);            \
    else                                                                                                              \
      hipLaunchKernelGGL((ea_eval_fold_kernel<double, P, N, false, true>), grid_f, dim3(N), shmem, stream, x0, y0, z0, n0, shape, \
                         chunks_per_xcd, probs, poses, partials, lds_texels, groups, prev_rows, prev_out);            \
  } while (0)
    if (nt == 1024) EA_LAUNCH_FI(1, 1024);
    else if (ppt == 1) EA_LAUNCH_FI(1, 256);
    else EA_LAUNCH_FI(2, 256);
#undef EA_LAUNCH_FI
    return hipGetLastError();
  }
#define EA_LAUNCH_F(T, P, N)                                                                                          \
  do {                                                                                                                \
    if (buffer_loads)                                                                                                 \
      hipLaunchKernelGGL((ea_eval_fold_kernel<T, P, N, true>), grid_f, dim3(N), shmem, stream, x0, y0, z0, n0, shape,  \
                         chunks_per_xcd, probs, poses, partials, lds_texels, groups, prev_rows, prev_out);            \
    else                                                                                                              \
      hipLaunchKernelGGL((ea_eval_fold_kernel<T, P, N, false>), grid_f, dim3(N), shmem, stream, x0, y0, z0, n0, shape, \
                         chunks_per_xcd, probs, poses, partials, lds_texels, groups, prev_rows, prev_out);            \
  } while (0)
  if (dtype == 1) {
    if (nt == 1024) { if (ppt == 1) EA_LAUNCH_F(float, 1, 1024); else if (ppt == 2) EA_LAUNCH_F(float, 2, 1024); else EA_LAUNCH_F(float, 4, 1024); }
    else if (ppt == 1) EA_LAUNCH_F(float, 1, 256);
    else if (ppt == 2) EA_LAUNCH_F(float, 2, 256);
    else EA_LAUNCH_F(float, 4, 256);
  } else {
    if (nt == 1024) EA_LAUNCH_F(double, 1, 1024);
    else if (ppt == 1) EA_LAUNCH_F(double, 1, 256);
    else EA_LAUNCH_F(double, 2, 256);
  }
#undef EA_LAUNCH_F
  return hipGetLastError();
}

// the stand-alone fold in the order the riding folds of an nt-thread evaluation use
hipError_t launch_reduce_nt(int nt, const GroupDesc *groups, int count, const double *partials, EvalOut *out,
                            hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  if (nt == 1024) hipLaunchKernelGGL(ea_reduce_kernel, dim3(count), dim3(kFoldThreads), 0, stream, groups, partials, out);
  else hipLaunchKernelGGL(ea_reduce256_kernel, dim3(count), dim3(kLmThreads), 0, stream, groups, partials, out);
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

// materialised mode: rows of every term of the batch into the caller's device arrays (ea_eval_rows_kernel)
hipError_t launch_eval_rows(int dtype, int variant, int buffer_loads, int img32, int layout, int staged, const ProblemDesc *probs, int nterms,
                            long long max_n, const PoseState *poses, int corrected, int nontemporal, long long total_rows,
                            void *r_out, void *J_out, unsigned int *n_invalid, hipStream_t stream) {
  if (nterms <= 0 || max_n <= 0) return hipSuccess;
  const int chunks = (int)((max_n + kBlockThreads - 1) / kBlockThreads);
  const int chunks_per_xcd = (chunks + 7) / 8;
  const dim3 grid(chunks_per_xcd * 8, nterms);
#define EA_ROWS(T, V, B, L, S)                                                                                              \
  hipLaunchKernelGGL((ea_eval_rows_kernel<T, V, B, L, S>), grid, dim3(kBlockThreads), 0, stream, probs, poses, chunks_per_xcd, \
                     corrected, nontemporal, total_rows, static_cast<T *>(r_out), static_cast<T *>(J_out), n_invalid)
#define EA_ROWS_L(T, V, B)                                                                    \
  do {                                                                                        \
    if (layout == 1) EA_ROWS(T, V, B, 1, false);                                              \
    else if (staged) EA_ROWS(T, V, B, 0, true);                                               \
    else EA_ROWS(T, V, B, 0, false);                                                          \
  } while (0)
  if (img32) {  // fp64 rows over the fp32-stored image
    if (dtype != 0 || variant) return hipErrorInvalidValue;
#define EA_ROWS_I(B)                                                                                                            \
  do {                                                                                                                          \
    if (layout == 1)                                                                                                            \
      hipLaunchKernelGGL((ea_eval_rows_kernel<double, false, B, 1, false, true>), grid, dim3(kBlockThreads), 0, stream, probs, poses, chunks_per_xcd, \
                         corrected, nontemporal, total_rows, static_cast<double *>(r_out), static_cast<double *>(J_out), n_invalid); \
    else if (staged)                                                                                                            \
      hipLaunchKernelGGL((ea_eval_rows_kernel<double, false, B, 0, true, true>), grid, dim3(kBlockThreads), 0, stream, probs, poses, chunks_per_xcd, \
                         corrected, nontemporal, total_rows, static_cast<double *>(r_out), static_cast<double *>(J_out), n_invalid); \
    else                                                                                                                        \
      hipLaunchKernelGGL((ea_eval_rows_kernel<double, false, B, 0, false, true>), grid, dim3(kBlockThreads), 0, stream, probs, poses, chunks_per_xcd, \
                         corrected, nontemporal, total_rows, static_cast<double *>(r_out), static_cast<double *>(J_out), n_invalid); \
  } while (0)
    if (buffer_loads) EA_ROWS_I(true); else EA_ROWS_I(false);
#undef EA_ROWS_I
  } else if (variant) {  // distortion / second-camera terms: flat addressing
    if (dtype == 1) EA_ROWS_L(float, true, false); else EA_ROWS_L(double, true, false);
  } else if (buffer_loads) {
    if (dtype == 1) EA_ROWS_L(float, false, true); else EA_ROWS_L(double, false, true);
  } else {
    if (dtype == 1) EA_ROWS_L(float, false, false); else EA_ROWS_L(double, false, false);
  }
#undef EA_ROWS_L
#undef EA_ROWS
  return hipGetLastError();
}

hipError_t launch_pixel_cost(int dtype, const ProblemDesc *probs, int problem, int n, const PoseState *poses, void *partials,
                             hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int grid = (n + kBlockThreads - 1) / kBlockThreads;
  if (dtype == 1)
    hipLaunchKernelGGL((ea_pixel_cost_kernel<float>), dim3(grid), dim3(kBlockThreads), 0, stream, probs, problem, poses,
                       static_cast<PixelCostPartial *>(partials));
  else
    hipLaunchKernelGGL((ea_pixel_cost_kernel<double>), dim3(grid), dim3(kBlockThreads), 0, stream, probs, problem, poses,
                       static_cast<PixelCostPartial *>(partials));
  return hipGetLastError();
}

hipError_t launch_reduce(const GroupDesc *groups, int count, const double *partials, EvalOut *out,
                         hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(ea_reduce_kernel, dim3(count), dim3(kFoldThreads), 0, stream, groups, partials, out);
  return hipGetLastError();
}

hipError_t launch_reduce_done(const GroupDesc *groups, int count, const double *partials, EvalOut *out, unsigned int *counter,
                              int *host_flag, int seq, hipStream_t stream) {
  if (count <= 0) return hipErrorInvalidValue;  // (somebody has to raise the flag)
  hipLaunchKernelGGL(ea_reduce_done_kernel, dim3(count), dim3(kFoldThreads), 0, stream, groups, partials, out, counter, host_flag, seq);
  return hipGetLastError();
}

hipError_t launch_lm_step(const GroupDesc *groups, int count, const double *partials, PoseState *poses,
                          LMState *states, LMCold *cold, LMTrace *traces, const LMOptions &opt, int *progress,
                          LMState *host_states, LMTrace *host_traces, const GroupDesc &first, int post_done, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  if (opt.strategy == 0)
    hipLaunchKernelGGL(ea_lm_step_kernel<0>, dim3(count), dim3(kLmThreads), 0, stream, groups, partials, poses,
                       states, cold, traces, opt, progress, host_states, host_traces, first, post_done);
  else
    hipLaunchKernelGGL(ea_lm_step_kernel<1>, dim3(count), dim3(kLmThreads), 0, stream, groups, partials, poses,
                       states, cold, traces, opt, progress, host_states, host_traces, first, post_done);
  return hipGetLastError();
}

// ea_lm_iter_kernel: launch j of a solve reads what launch j - 1 wrote (state, cold system, rows) and writes the other
// buffer of each pair.  The grid is one workgroup larger than the evaluation's (the writer), rounded to the 8 XCDs.
hipError_t launch_lm_iter(int dtype, int ppt, const ProblemDesc *probs, int count, int chunk, int max_chunks, int xcd_remap,
                          PoseState *poses, const double *rows_in, double *rows_out, int buffer_loads, int img32,
                          const void *x0, const void *y0, const void *z0, int n0, const GroupDesc *groups,
                          const LMState *st_in, LMState *st_out, const LMCold *cold_in, LMCold *cold_out, LMTrace *traces,
                          const LMOptions &opt, int *progress, LMState *host_states, LMTrace *host_traces,
                          const GroupDesc &first, int post_done, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  if (chunk <= 0 || chunk > 0xffff || chunk != kLmThreads * ppt) return hipErrorInvalidValue;
  const int chunks_per_xcd = (max_chunks + 1 + 7) / 8;
  const dim3 grid(xcd_remap ? chunks_per_xcd * 8 : max_chunks + 1, count);
  const int shape = chunk | ((xcd_remap ? 1 : 0) << 16) | (1 << 17);
  const size_t shmem = (size_t)kHdrBytes;
#define EA_ITER_S(T, P, B, I, S)                                                                                           \
  hipLaunchKernelGGL((ea_lm_iter_kernel<T, P, B, I, S>), grid, dim3(kLmThreads), shmem, stream, x0, y0, z0, n0, shape,     \
                     chunks_per_xcd, rows_in, first.tile_begin, first.tile_end, probs, poses, rows_out, groups, st_in, st_out, \
                     cold_in, cold_out, traces, opt, progress, host_states, host_traces, post_done)
#define EA_ITER(T, P, B, I) do { if (opt.strategy == 0) EA_ITER_S(T, P, B, I, 0); else EA_ITER_S(T, P, B, I, 1); } while (0)
#define EA_ITER_B(T, P, I) do { if (buffer_loads) EA_ITER(T, P, true, I); else EA_ITER(T, P, false, I); } while (0)
  if (dtype == 1) {
    if (img32) return hipErrorInvalidValue;
    if (ppt == 1) EA_ITER_B(float, 1, false); else if (ppt == 2) EA_ITER_B(float, 2, false); else if (ppt == 4) EA_ITER_B(float, 4, false);
    else return hipErrorInvalidValue;
  } else if (img32) {
    if (ppt == 1) EA_ITER_B(double, 1, true); else if (ppt == 2) EA_ITER_B(double, 2, true); else return hipErrorInvalidValue;
  } else {
    if (ppt == 1) EA_ITER_B(double, 1, false); else if (ppt == 2) EA_ITER_B(double, 2, false); else return hipErrorInvalidValue;
  }
#undef EA_ITER_B
#undef EA_ITER
#undef EA_ITER_S
  return hipGetLastError();
}

#ifdef EA_STAMPS
hipError_t set_stamp_buffer(unsigned long long *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)); }
hipError_t set_lm_stamp_buffer(unsigned long long *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_lm_stamp_buf), &buf, sizeof(buf)); }
#endif

hipError_t launch_selftest_reduce(const float *in, float *a, float *b, float *c, float *d, double *o32, double *o64,
                                  hipStream_t stream) {
  hipLaunchKernelGGL(ea_selftest_reduce_kernel, dim3(1), dim3(64), 0, stream, in, a, b, c, d, o32, o64);
  return hipGetLastError();
}

__global__ void ea_empty_kernel() {}
hipError_t launch_empty(int grid, int block, hipStream_t stream) {
  hipLaunchKernelGGL(ea_empty_kernel, dim3(grid), dim3(block), 0, stream);
  return hipGetLastError();
}

hipError_t launch_make_poses(const double *qt, int n, int count, const ProblemDesc *probs, const GroupDesc *groups,
                             PoseState *out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ea_make_poses_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, qt, n, count, probs, groups, out);
  return hipGetLastError();
}

hipError_t launch_pad_image(int dtype, const void *src, int H, int W, void *dst, int pitch, float *dst32, int *inexact,
                            hipStream_t stream) {
  dim3 block(256), grid((W + 2 * kImagePad + 255) / 256, H + 2 * kImagePad);
  if (dtype == 1)
    hipLaunchKernelGGL((ea_pad_image_kernel<float>), grid, block, 0, stream, (const float *)src, H, W, (float *)dst, pitch, nullptr, nullptr);
  else
    hipLaunchKernelGGL((ea_pad_image_kernel<double>), grid, block, 0, stream, (const double *)src, H, W, (double *)dst, pitch, dst32, inexact);
  return hipGetLastError();
}

hipError_t launch_aos_to_soa(int dtype, const double *src, long long n, int stride, void *x, void *y, void *z, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == 1) hipLaunchKernelGGL((ea_aos_to_soa_kernel<float>), grid, dim3(256), 0, stream, src, n, stride, (float *)x, (float *)y, (float *)z);
  else hipLaunchKernelGGL((ea_aos_to_soa_kernel<double>), grid, dim3(256), 0, stream, src, n, stride, (double *)x, (double *)y, (double *)z);
  return hipGetLastError();
}

hipError_t launch_grid_to_image(int dtype, const double *grid, int W, int H, void *dst, int pitch, float *dst32, int *inexact,
                                hipStream_t stream) {
  dim3 block(256), tiles((W + 2 * kImagePad + 31) / 32, (H + 2 * kImagePad + 31) / 32);
  if (dtype == 1) hipLaunchKernelGGL((ea_grid_to_image_kernel<float>), tiles, block, 0, stream, grid, W, H, (float *)dst, pitch, nullptr, nullptr);
  else hipLaunchKernelGGL((ea_grid_to_image_kernel<double>), tiles, block, 0, stream, grid, W, H, (double *)dst, pitch, dst32, inexact);
  return hipGetLastError();
}

#endif  // EA_TU_VARIANT
#undef EA_LAUNCH_PROLOGUE
#undef EA_LAUNCH
#undef EA_LAUNCH_B

}  // namespace ea

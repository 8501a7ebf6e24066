// ea_preprocess.hip — the producers either side of the hot path, on the GPU:
//   edge-point extractor  = get_aX                  ref: standalone/utils.cpp:201-281
//   DT image producer     = get_distance_transform  ref: standalone/utils.cpp:38-83
//   Canny flavour         = get_distance_transform2* / get_aX_canny   ref: utils.cpp:85-199, :371-462
//   ROS flavour           = SolveEA::setRefFrame / setNowFrame        ref: src/SolveEA.cpp:29-119
// so that a frame pair goes from raw images to a solved pose without the CPU touching a pixel.
//
// All of it is 8/16/32-bit integer work plus one float scaling — bit-exact against
// oracle/preprocess_np.py (tests/test_gpu_preprocess.py).  OpenCV semantics restated:
//   GaussianBlur 3x3 sigma 0 (8-bit): ([1 2 1]x[1 2 1] + 8) >> 4, reflect-101 border
//   cvtColor RGB2GRAY (8-bit):       (4899 c0 + 9617 c1 + 1868 c2 + 8192) >> 14
//   Laplacian CV_16S ksize 3:        [[2,0,2],[0,-8,0],[2,0,2]], reflect-101; convertScaleAbs = min(|x|,255)
//   medianBlur 3 on a {0,255} image: majority of the 3x3 window, replicate border
//   distanceTransform(DIST_L2, 3):   3x3 chamfer, a = 0.955, b = 1.3693 in 16.16 fixed point
//   normalize(NORM_MINMAX, lo, hi):  float32  src * scale + shift
//   blur 3x3 (8-bit):                (sum of nine + 4) / 9, reflect-101 border
//   Canny(gray, t1, t2), aperture 3, L1 gradient: Sobel CV_16S with replicated border, |dx| + |dy|,
//                                    non-maximum suppression by the fixed-point tan(22.5 deg) test
//                                    (TG22 = 13573, shift 15; magnitudes outside the image are 0),
//                                    hysteresis over the 8-neighbourhood
//   Canny(bgr, 150, 100, 3, true):   per pixel the channel with the largest dx^2 + dy^2 (first on ties), squared
//                                    thresholds, the same suppression and hysteresis
//   distanceTransform(DIST_L2, DIST_MASK_PRECISE): float32 sqrt((float)dx^2 + (float)dy^2) of the nearest zero pixel
//
// The two-pass raster chamfer of OpenCV is inherently sequential; what it computes is the exact
// shortest 8-connected path length, which has the closed form  a*max(dx,dy) + (b-a)*min(dx,dy).
// For a fixed column x' the nearest feature row minimises it, so
//   DT(x,y) = min_x'  f(|x-x'|, G(x',y)),   G = per-column distance to the nearest feature row,
// which is embarrassingly parallel: a column scan for G, then a bounded search along each row.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ea_types.h"

namespace ea {

constexpr int kChamferA = 62587;   // CV_FLT_TO_FIX(0.955f, 16)
constexpr int kChamferB = 89738;   // CV_FLT_TO_FIX(1.3693f, 16)
constexpr int kChamferBig = 0x3fffffff;
constexpr int kNoFeature = 1 << 28;

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return min(max(i, 0), n - 1);
}

// blur (3 channels) + gray in one pass
__global__ void ea_blur_gray_kernel(const uint8_t *__restrict__ bgr, int H, int W, uint8_t *__restrict__ gray) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  int acc[3] = {0, 0, 0};
  const int wv[3] = {1, 2, 1};
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = reflect101(v + dy, H);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = reflect101(u + dx, W);
      const uint8_t *p = bgr + ((size_t)yy * W + xx) * 3;
      const int w = wv[dy + 1] * wv[dx + 1];
      acc[0] += w * p[0]; acc[1] += w * p[1]; acc[2] += w * p[2];
    }
  }
  const int c0 = (acc[0] + 8) >> 4, c1 = (acc[1] + 8) >> 4, c2 = (acc[2] + 8) >> 4;
  gray[(size_t)v * W + u] = (uint8_t)((4899 * c0 + 9617 * c1 + 1868 * c2 + 8192) >> 14);
}

__global__ void ea_laplacian_abs_kernel(const uint8_t *__restrict__ gray, int H, int W, uint8_t *__restrict__ lap) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  const int ym = reflect101(v - 1, H), yp = reflect101(v + 1, H);
  const int xm = reflect101(u - 1, W), xp = reflect101(u + 1, W);
  const int s = 2 * ((int)gray[(size_t)ym * W + xm] + gray[(size_t)ym * W + xp] + gray[(size_t)yp * W + xm] +
                     gray[(size_t)yp * W + xp]) - 8 * (int)gray[(size_t)v * W + u];
  lap[(size_t)v * W + u] = (uint8_t)min(abs(s), 255);
}

// B = (lap > thr) ? 0 : 255, then 3x3 median with replicate border; on a two-valued image the
// median is the majority.  mask: 0 = edge (DT source), 255 = background.
__global__ void ea_threshold_median_kernel(const uint8_t *__restrict__ lap, int H, int W, int thr, int median,
                                           uint8_t *__restrict__ mask) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  if (!median) {
    mask[(size_t)v * W + u] = lap[(size_t)v * W + u] > thr ? 0 : 255;
    return;
  }
  int bg = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = min(max(v + dy, 0), H - 1);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = min(max(u + dx, 0), W - 1);
      bg += lap[(size_t)yy * W + xx] > thr ? 0 : 1;
    }
  }
  mask[(size_t)v * W + u] = bg >= 5 ? 255 : 0;
}

// ---- Canny flavour (ref: utils.cpp:87-94 / :397-404: blur 3x3 -> CV_RGB2GRAY -> Canny(30, 90)) ----------------

// box blur (3 channels) + gray in one pass
__global__ void ea_boxblur_gray_kernel(const uint8_t *__restrict__ bgr, int H, int W, uint8_t *__restrict__ gray) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = reflect101(v + dy, H);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = reflect101(u + dx, W);
      const uint8_t *p = bgr + ((size_t)yy * W + xx) * 3;
      acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
    }
  }
  const int c0 = (acc[0] + 4) / 9, c1 = (acc[1] + 4) / 9, c2 = (acc[2] + 4) / 9;
  gray[(size_t)v * W + u] = (uint8_t)((4899 * c0 + 9617 * c1 + 1868 * c2 + 8192) >> 14);
}

// Sobel 3x3 (replicated border) -> L1 magnitude (int16 range) and the packed direction class needed by the
// suppression test: bits 0-1 = 0 horizontal / 1 vertical / 2 diagonal, bit 2 = sign s of the diagonal (1: s = -1)
__global__ void ea_sobel_mag_kernel(const uint8_t *__restrict__ gray, int H, int W, int *__restrict__ mag,
                                    uint8_t *__restrict__ dir) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  const int ym = max(v - 1, 0), yp = min(v + 1, H - 1), xm = max(u - 1, 0), xp = min(u + 1, W - 1);
  const int a00 = gray[(size_t)ym * W + xm], a01 = gray[(size_t)ym * W + u], a02 = gray[(size_t)ym * W + xp];
  const int a10 = gray[(size_t)v * W + xm], a12 = gray[(size_t)v * W + xp];
  const int a20 = gray[(size_t)yp * W + xm], a21 = gray[(size_t)yp * W + u], a22 = gray[(size_t)yp * W + xp];
  const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
  const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
  mag[(size_t)v * W + u] = abs(dx) + abs(dy);
  const long long x = abs(dx), y = (long long)abs(dy) << 15;
  const long long tg22x = x * 13573;
  int cls;
  if (y < tg22x) cls = 0;
  else if (y > tg22x + (x << 16)) cls = 1;
  else cls = 2 | (((dx ^ dy) < 0) ? 4 : 0);
  dir[(size_t)v * W + u] = (uint8_t)cls;
}

// ROS flavour (ref: src/SolveEA.cpp:46, :102): cv::Canny on the 3-channel image with L2gradient = true -- the channel
// with the largest dx^2 + dy^2 (the first one on ties) supplies dx, dy; the magnitude is that sum of squares
__global__ void ea_sobel_mag_l2_bgr_kernel(const uint8_t *__restrict__ bgr, int H, int W, int *__restrict__ mag,
                                           uint8_t *__restrict__ dir) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  const int ym = max(v - 1, 0), yp = min(v + 1, H - 1), xm = max(u - 1, 0), xp = min(u + 1, W - 1);
  int bdx = 0, bdy = 0, bm = -1;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    auto P = [&](int yy, int xx) { return (int)bgr[((size_t)yy * W + xx) * 3 + c]; };
    const int dx = (P(ym, xp) + 2 * P(v, xp) + P(yp, xp)) - (P(ym, xm) + 2 * P(v, xm) + P(yp, xm));
    const int dy = (P(yp, xm) + 2 * P(yp, u) + P(yp, xp)) - (P(ym, xm) + 2 * P(ym, u) + P(ym, xp));
    const int m = dx * dx + dy * dy;
    if (m > bm) { bm = m; bdx = dx; bdy = dy; }
  }
  mag[(size_t)v * W + u] = bm;
  const long long x = abs(bdx), y = (long long)abs(bdy) << 15;
  const long long tg22x = x * 13573;
  int cls;
  if (y < tg22x) cls = 0;
  else if (y > tg22x + (x << 16)) cls = 1;
  else cls = 2 | (((bdx ^ bdy) < 0) ? 4 : 0);
  dir[(size_t)v * W + u] = (uint8_t)cls;
}

// non-maximum suppression + double threshold: label 2 = strong, 0 = candidate, 1 = no edge
__global__ void ea_canny_nms_kernel(const int *__restrict__ mag, const uint8_t *__restrict__ dir, int H, int W, int low,
                                    int high, uint8_t *__restrict__ label, int *__restrict__ flags, int nflags) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < nflags) flags[threadIdx.x] = 0;  // hysteresis change flags
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  auto M = [&](int yy, int xx) { return (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : mag[(size_t)yy * W + xx]; };
  const int m = mag[(size_t)v * W + u];
  uint8_t out = 1;
  if (m > low) {
    const int cls = dir[(size_t)v * W + u];
    bool is_max;
    if ((cls & 3) == 0) is_max = m > M(v, u - 1) && m >= M(v, u + 1);
    else if ((cls & 3) == 1) is_max = m > M(v - 1, u) && m >= M(v + 1, u);
    else {
      const int s = (cls & 4) ? -1 : 1;
      is_max = m > M(v - 1, u - s) && m > M(v + 1, u + s);
    }
    if (is_max) out = m > high ? 2 : 0;
  }
  label[(size_t)v * W + u] = out;
}

// hysteresis: candidates 8-connected to a strong pixel become strong.  One WAVEFRONT owns a tile of 62 x 62 pixels
// plus a one-pixel halo as bit masks: lane = image row, bit = image column, one 64-bit word of strong pixels and one of
// candidates per lane.  A sweep takes the strong pixels of the rows above and below (two lane shifts), widens them by
// one column either way, and floods the result along the row's runs of candidates with a Kogge-Stone fill (six
// shift-and-mask steps per direction) -- so a sweep carries "strong" across a whole horizontal run and one row up or
// down, with no LDS and no barrier, where a per-pixel update needs one pass (and one barrier) per pixel of the chain.
// Tiles exchange through global memory between launches; the host repeats the launch until no tile reports a change.
constexpr int kHystTile = 62;
constexpr int kHystBatch = 8;  // launches enqueued per look at the change flags
__device__ __forceinline__ unsigned long long hyst_fill(unsigned long long seed, unsigned long long cand) {
  unsigned long long g = seed, pr = cand;  // towards higher bits
  g |= pr & (g << 1);  pr &= pr << 1;
  g |= pr & (g << 2);  pr &= pr << 2;
  g |= pr & (g << 4);  pr &= pr << 4;
  g |= pr & (g << 8);  pr &= pr << 8;
  g |= pr & (g << 16); pr &= pr << 16;
  g |= pr & (g << 32);
  unsigned long long h = seed, pl = cand;  // towards lower bits
  h |= pl & (h >> 1);  pl &= pl >> 1;
  h |= pl & (h >> 2);  pl &= pl >> 2;
  h |= pl & (h >> 4);  pl &= pl >> 4;
  h |= pl & (h >> 8);  pl &= pl >> 8;
  h |= pl & (h >> 16); pl &= pl >> 16;
  h |= pl & (h >> 32);
  return g | h;
}

__global__ __launch_bounds__(256) void ea_canny_hysteresis_kernel(uint8_t *__restrict__ label, int H, int W, int tiles_x,
                                                                  int tiles, const int *__restrict__ prev_changed,
                                                                  int *__restrict__ changed) {
  // launches are enqueued in batches without a host look in between: once a launch has changed nothing the edge set
  // is final and the launches queued behind it have nothing to do
  if (prev_changed && *prev_changed == 0) return;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave-uniform
  if (tile >= tiles) return;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * kHystTile - 1, y0 = ty * kHystTile - 1;  // image position of bit 0 / lane 0 (the halo)
  const int x = x0 + lane;
  const bool x_in = x >= 0 && x < W;
  // row r of the tile -> lane r: every lane reads one byte of the row, two ballots make the row's masks.  All 64 loads
  // are issued before the first ballot (one memory round trip for the tile instead of 64 dependent ones).
  uint8_t v[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int y = min(max(y0 + r, 0), H - 1);  // (uniform; rows outside the image are masked below)
    v[r] = x_in ? label[(size_t)y * W + x] : (uint8_t)1;
  }
  unsigned long long S = 0, C = 0;
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int y = y0 + r;
    const int vr = (y >= 0 && y < H) ? (int)v[r] : 1;
    const unsigned long long sm = __ballot(vr == 2), cm = __ballot(vr == 0);
    if (lane == r) { S = sm; C = cm; }
  }
  if (!__any(C != 0)) return;  // no candidate in the tile: nothing can change
  const unsigned long long S0 = S;
  for (;;) {
    unsigned long long up = __shfl_up(S, 1), dn = __shfl_down(S, 1);
    if (lane == 0) up = 0;
    if (lane == 63) dn = 0;
    const unsigned long long n = up | dn | S;
    const unsigned long long seed = (n | (n << 1) | (n >> 1)) & C;
    const unsigned long long grown = hyst_fill(seed, C);
    const bool ch = grown != 0;
    S |= grown;
    C &= ~grown;
    if (!__any(ch)) break;
  }
  const unsigned long long D = S & ~S0;  // what this launch made strong, halo included (a neighbour's pixel: same answer)
  if (!__any(D != 0)) return;
  for (int r = 0; r < 64; ++r) {
    const unsigned long long d = __shfl(D, r);  // (uniform)
    if (d == 0) continue;
    const int y = y0 + r;
    if (((d >> lane) & 1ull) && x_in && y >= 0 && y < H) label[(size_t)y * W + x] = 2;
  }
  if (lane == 0) atomicOr(changed, 1);
}

// labels -> the CV_8U edge map (255 on edges) [AND the optional mask: inputmask > 1] and the DT source mask
// (0 on edges, 255 elsewhere = `255 - edges`)
__global__ void ea_canny_finish_kernel(const uint8_t *__restrict__ label, const uint8_t *__restrict__ keep /*nullable*/,
                                       int npix, uint8_t *__restrict__ edges, uint8_t *__restrict__ inv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const bool e = label[i] == 2 && (!keep || keep[i] > 1);
  edges[i] = e ? 255 : 0;
  inv[i] = e ? 0 : 255;
}

// G(x,y) = |y - y'| to the nearest feature pixel (mask == 0) of column x; kNoFeature if none.
// A column scan is a recurrence along y, so it is cut into segments of kSegRows rows that are scanned
// independently (kernel A: local distances + what each segment sees at its two ends), a short per-column
// pass chains the segment ends (kernel B: carries), and the consumer applies the carries on the fly.
constexpr int kSegRows = 32;

__global__ void ea_column_local_kernel(const uint8_t *__restrict__ mask, int H, int W, int *__restrict__ G,
                                       int *__restrict__ end_dn /*S x W*/, int *__restrict__ end_up /*S x W*/,
                                       unsigned int *__restrict__ minmax /* reset here for the row pass two launches on */) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { minmax[0] = 0xffffffffu; minmax[1] = 0u; }
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  const int s = blockIdx.y, y0 = s * kSegRows, y1 = min(H, y0 + kSegRows);
  uint8_t m[kSegRows];
#pragma unroll
  for (int k = 0; k < kSegRows; ++k) m[k] = (y0 + k < y1) ? mask[(size_t)(y0 + k) * W + x] : (uint8_t)255;
  int dn[kSegRows];
  int d = kNoFeature;
#pragma unroll
  for (int k = 0; k < kSegRows; ++k) {
    d = m[k] == 0 ? 0 : min(d + 1, kNoFeature);
    dn[k] = d;
  }
  // distance from the segment's last row to the nearest feature above, inside the segment
  int last = kNoFeature;
#pragma unroll
  for (int k = 0; k < kSegRows; ++k)
    if (y0 + k == y1 - 1) last = dn[k];
  end_dn[(size_t)s * W + x] = last;
  d = kNoFeature;
#pragma unroll
  for (int k = kSegRows - 1; k >= 0; --k) {
    if (y0 + k < y1) {
      d = m[k] == 0 ? 0 : min(d + 1, kNoFeature);
      G[(size_t)(y0 + k) * W + x] = min(dn[k], d);
    }
  }
  end_up[(size_t)s * W + x] = d;  // distance from the segment's first row to the nearest feature below
}

// carry_dn[s] = distance from the row just above segment s to the nearest feature above it;
// carry_up[s] = distance from the row just below segment s to the nearest feature below it
__global__ void ea_column_carry_kernel(const int *__restrict__ end_dn, const int *__restrict__ end_up, int H, int W,
                                       int S, int *__restrict__ carry_dn, int *__restrict__ carry_up) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  int c = kNoFeature;
  for (int s = 0; s < S; ++s) {
    carry_dn[(size_t)s * W + x] = c;
    const int len = min(H, (s + 1) * kSegRows) - s * kSegRows;
    c = min(end_dn[(size_t)s * W + x], min(c + len, kNoFeature));
  }
  c = kNoFeature;
  for (int s = S - 1; s >= 0; --s) {
    carry_up[(size_t)s * W + x] = c;
    const int len = min(H, (s + 1) * kSegRows) - s * kSegRows;
    c = min(end_up[(size_t)s * W + x], min(c + len, kNoFeature));
  }
}

// min / max of a non-negative float over the workgroup -> two atomics per workgroup (one per LANE, all on the same two
// words, made the row pass an atomic-throughput kernel: 51 us of which ~40 were the 246 000 same-address atomics)
__device__ __forceinline__ void block_minmax_atomic(float lmin, float lmax, unsigned int *minmax) {
  __shared__ float s_mn[16], s_mx[16];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lmin = fminf(lmin, __shfl_xor(lmin, off));
    lmax = fmaxf(lmax, __shfl_xor(lmax, off));
  }
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { s_mn[wave] = lmin; s_mx[wave] = lmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) { lmin = fminf(lmin, s_mn[w]); lmax = fmaxf(lmax, s_mx[w]); }
    // floats >= 0: the bit pattern orders like the value
    atomicMin(&minmax[0], __float_as_uint(lmin));
    atomicMax(&minmax[1], __float_as_uint(lmax));
  }
}

__device__ __forceinline__ int chamfer_cost(int dx, int dy) {
  const int mx = max(dx, dy), mn = min(dx, dy);
  return kChamferA * mx + (kChamferB - kChamferA) * mn;
}

// One workgroup per image row: G row staged in LDS, every lane searches outwards from its own
// column and stops as soon as a*dx alone exceeds the best distance found.
__global__ void ea_chamfer_row_kernel(const int *__restrict__ G, const int *__restrict__ carry_dn,
                                      const int *__restrict__ carry_up, int H, int W, int *__restrict__ dist_fix,
                                      unsigned int *__restrict__ minmax /* [0]=min bits, [1]=max bits of dist*2^-16 as float */) {
  extern __shared__ int s_g[];
  const int y = blockIdx.x;
  const int seg = y / kSegRows, y0 = seg * kSegRows, y1 = min(H, y0 + kSegRows);
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    int g = G[(size_t)y * W + x];
    const int cd = carry_dn[(size_t)seg * W + x], cu = carry_up[(size_t)seg * W + x];
    if (cd < kNoFeature) g = min(g, cd + (y - y0 + 1));
    if (cu < kNoFeature) g = min(g, cu + (y1 - y));
    s_g[x] = g;
  }
  __syncthreads();
  float lmin = 3.0e38f, lmax = 0.0f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    int best = kChamferBig;
    {
      const int g = s_g[x];
      if (g < kNoFeature) best = min(best, kChamferA * g);
    }
    // two offsets per trip: the four LDS reads are issued together (a candidate past the bound a*dx >= best cannot
    // lower the minimum, so looking at it is harmless); one at a time the loop is a chain of LDS round trips
    for (int dx = 1; dx < W; dx += 2) {
      if ((long long)kChamferA * dx >= best) break;
      if (x - dx < 0 && x + dx >= W) break;
      int gl[2], gr[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int xl = x - dx - k, xr = x + dx + k;
        gl[k] = xl >= 0 ? s_g[xl] : kNoFeature;
        gr[k] = xr < W ? s_g[xr] : kNoFeature;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (gl[k] < kNoFeature) best = min(best, chamfer_cost(dx + k, gl[k]));
        if (gr[k] < kNoFeature) best = min(best, chamfer_cost(dx + k, gr[k]));
      }
    }
    dist_fix[(size_t)y * W + x] = best;
    const float f = (float)((double)best * (1.0 / 65536.0));
    lmin = fminf(lmin, f); lmax = fmaxf(lmax, f);
  }
  block_minmax_atomic(lmin, lmax, minmax);
}

// Exact Euclidean variant of the row pass (DIST_MASK_PRECISE): the same per-column distances G, squared cost
// dx^2 + G(x')^2, search outwards until dx^2 alone reaches the best.  The result is formed like OpenCV's row pass:
// sqrt((float)dx^2 + (float)dy^2) in float32.  dist_f32 takes the place of the fixed-point chamfer distance.
__global__ void ea_edt_row_kernel(const int *__restrict__ G, const int *__restrict__ carry_dn,
                                  const int *__restrict__ carry_up, int H, int W, float *__restrict__ dist_f32,
                                  unsigned int *__restrict__ minmax) {
  extern __shared__ int s_g[];
  const int y = blockIdx.x;
  const int seg = y / kSegRows, y0 = seg * kSegRows, y1 = min(H, y0 + kSegRows);
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    int g = G[(size_t)y * W + x];
    const int cd = carry_dn[(size_t)seg * W + x], cu = carry_up[(size_t)seg * W + x];
    if (cd < kNoFeature) g = min(g, cd + (y - y0 + 1));
    if (cu < kNoFeature) g = min(g, cu + (y1 - y));
    s_g[x] = g;
  }
  __syncthreads();
  float lmin = 3.0e38f, lmax = 0.0f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    long long best = 1ll << 40;
    int bdx = 0, bg = 0;
    {
      const int g = s_g[x];
      if (g < kNoFeature) { best = (long long)g * g; bg = g; }
    }
    // two offsets per trip, reads issued together; the candidates are still examined in the order dx ascending, left
    // before right, with the strict comparison: the winning (dx, dy) pair -- and with it the float32 rounding of the
    // result -- is the one the one-at-a-time search finds (a candidate with dx^2 >= best fails the comparison anyway)
    for (int dx = 1; dx < W; dx += 2) {
      if ((long long)dx * dx >= best) break;
      if (x - dx < 0 && x + dx >= W) break;
      int gl[2], gr[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int xl = x - dx - k, xr = x + dx + k;
        gl[k] = xl >= 0 ? s_g[xl] : kNoFeature;
        gr[k] = xr < W ? s_g[xr] : kNoFeature;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const long long d2 = (long long)(dx + k) * (dx + k);
        if (gl[k] < kNoFeature && d2 + (long long)gl[k] * gl[k] < best) { best = d2 + (long long)gl[k] * gl[k]; bdx = dx + k; bg = gl[k]; }
        if (gr[k] < kNoFeature && d2 + (long long)gr[k] * gr[k] < best) { best = d2 + (long long)gr[k] * gr[k]; bdx = dx + k; bg = gr[k]; }
      }
    }
    const float f = sqrtf(__fadd_rn((float)((long long)bdx * bdx), (float)((long long)bg * bg)));  // correctly rounded (HIP default); __fsqrt_rn is the native approximation
    dist_f32[(size_t)y * W + x] = f;
    lmin = fminf(lmin, f); lmax = fmaxf(lmax, f);
  }
  block_minmax_atomic(lmin, lmax, minmax);
}

// dist (16.16 fixed) -> float32 [-> min-max normalised] -> padded image of the problem dtype
template <typename T>
__global__ void ea_dt_store_kernel(const int *__restrict__ dist_fix, const float *__restrict__ dist_f32 /* one of the two */,
                                   int H, int W, const unsigned int *__restrict__ minmax, int normalize, double lo, double hi,
                                   T *__restrict__ dst, int pitch, float *__restrict__ plain /*nullable HxW*/,
                                   float *__restrict__ dst32 /*nullable: the float32 mirror of an fp64 problem's image, same pitch*/) {
  const int pu = blockIdx.x * blockDim.x + threadIdx.x, pv = blockIdx.y;  // padded coordinates
  if (pu >= W + 2 * kImagePad) return;
  const int u = min(max(pu - kImagePad, 0), W - 1), v = min(max(pv - kImagePad, 0), H - 1);
  float f = dist_f32 ? dist_f32[(size_t)v * W + u] : (float)((double)dist_fix[(size_t)v * W + u] * (1.0 / 65536.0));
  if (normalize) {
    // cv::normalize(NORM_MINMAX, lo, hi) on CV_32F: scale/shift in double, applied in float
    const double smin = (double)__uint_as_float(minmax[0]), smax = (double)__uint_as_float(minmax[1]);
    const double scale = (hi - lo) * ((smax - smin) > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
    const double shift = lo - smin * scale;
    f = __fadd_rn(__fmul_rn(f, (float)scale), (float)shift);
  }
  dst[(size_t)pv * pitch + pu] = (T)f;
  if (dst32) dst32[(size_t)pv * pitch + pu] = f;  // (the value IS a float: the mirror is exact by construction)
  if (plain && pu >= kImagePad && pu < W + kImagePad && pv >= kImagePad && pv < H + kImagePad)
    plain[(size_t)(pv - kImagePad) * W + (pu - kImagePad)] = f;
}

// get_aX_mask (ref: utils.cpp:283-369): a point also needs mask > 0 -- the gradient map is zeroed elsewhere
__global__ void ea_gate_by_mask_kernel(uint8_t *__restrict__ grad, const uint8_t *__restrict__ mask, int npix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix && mask[i] == 0) grad[i] = 0;
}

// ---- edge points: flag, count per block, scan, scatter (raster order preserved) ----------------

constexpr int kScanBlock = 1024;

__device__ __forceinline__ bool edge_flag(const uint8_t *lap, const uint16_t *depth, size_t i, int thr) {
  // ref: utils.cpp:258-262  grad > threshold && Z > 0; depth == nullptr: the ROS flavour keeps every edge pixel
  return lap[i] > thr && (!depth || depth[i] > 0);
}

__global__ void ea_edge_count_kernel(const uint8_t *__restrict__ lap, const uint16_t *__restrict__ depth, int npix,
                                     int thr, int *__restrict__ block_counts) {
  __shared__ int s_cnt[kScanBlock / 64];
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  const bool f = i < npix && edge_flag(lap, depth, (size_t)i, thr);
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) s += s_cnt[w];
    block_counts[blockIdx.x] = s;
  }
}

// exclusive scan of the block counts by one workgroup (sequential over chunks of 1024)
__global__ void ea_edge_scan_kernel(int *__restrict__ block_counts, int nblocks, int *__restrict__ total) {
  __shared__ int s_buf[kScanBlock];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += kScanBlock) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_counts[i] : 0;
    s_buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {  // Hillis-Steele inclusive scan
      const int t = threadIdx.x >= off ? s_buf[threadIdx.x - off] : 0;
      __syncthreads();
      s_buf[threadIdx.x] += t;
      __syncthreads();
    }
    const int incl = s_buf[threadIdx.x], carry = s_carry;
    if (i < nblocks) block_counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == kScanBlock - 1) s_carry = carry + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}

template <typename T>
__global__ void ea_edge_scatter_kernel(const uint8_t *__restrict__ lap, const uint16_t *__restrict__ depth, int H, int W,
                                       int thr, const int *__restrict__ block_offsets, double fx, double fy, double cx,
                                       double cy, double z_scaling, T *__restrict__ X, T *__restrict__ Y, T *__restrict__ Z,
                                       int capacity) {
  __shared__ int s_cnt[kScanBlock / 64];
  const int npix = H * W;
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  const bool f = i < npix && edge_flag(lap, depth, (size_t)i, thr);
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = __popcll(m);
  __syncthreads();
  if (!f) return;
  int off = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += s_cnt[w];
  off += __popcll(m & ((1ull << lane) - 1ull));
  if (off >= capacity) return;
  const int v = i / W, u = i - v * W;
  // ref: utils.cpp:238-240 — Z = depth / factor; X = (u - cx) * Z / fx; Y = (v - cy) * Z / fy  (double)
  const double z = (double)depth[i] / z_scaling;
  const double x = __dmul_rn((double)u - cx, z) / fx;
  const double y = __dmul_rn((double)v - cy, z) / fy;
  X[off] = (T)x; Y[off] = (T)y; Z[off] = (T)z;
}

// ROS flavour of the scatter (ref: src/SolveEA.cpp:61-78): every edge pixel, float depth in metres, Z == 0 -> 1.0,
// X = Z * (xx - cx) / fx
template <typename T>
__global__ void ea_edge_scatter_ros_kernel(const uint8_t *__restrict__ edges, const float *__restrict__ depth, int H, int W,
                                           const int *__restrict__ block_offsets, double fx, double fy, double cx, double cy,
                                           T *__restrict__ X, T *__restrict__ Y, T *__restrict__ Z, int capacity) {
  __shared__ int s_cnt[kScanBlock / 64];
  const int npix = H * W;
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  const bool f = i < npix && edges[i] > 0;
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = __popcll(m);
  __syncthreads();
  if (!f) return;
  int off = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += s_cnt[w];
  off += __popcll(m & ((1ull << lane) - 1ull));
  if (off >= capacity) return;
  const int v = i / W, u = i - v * W;
  double z = (double)depth[i];
  z = (z == 0.0) ? 1.0 : z;
  const double x = __dmul_rn(z, (double)u - cx) / fx;
  const double y = __dmul_rn(z, (double)v - cy) / fy;
  X[off] = (T)x; Y[off] = (T)y; Z[off] = (T)z;
}

// ---- half-resolution level: cv::resize(src, dst, cv::Size(), 0.5, 0.5) as the ROS callbacks apply it to the bgr8 colour
// frame and to the float depth frame after NaN -> 0 (ref: src/ea.cpp:38, :56-62).  At an exact factor of two OpenCV's
// INTER_LINEAR resize is the 2 x 2 area mean (cv::resize switches to its fast area path; the bilinear weights are 1/2, 1/2
// anyway): 8-bit channels (a + b + c + d + 2) >> 2; float 0.25f * ((a + b) + (c + d)), the order of OpenCV 3's vector
// path for CV_32F (ResizeAreaFastVec_SIMD_32f: the two pixels of each source row are added first, then the rows) -- what
// an x86 or NEON build executes for every pixel of a width that is a multiple of four; the scalar generic loop would
// sum ((a + b) + c) + d, which differs in the last bit on some pixels.
__global__ void ea_resize_half_bgr8_kernel(const uint8_t *__restrict__ src, int H2, int W2, uint8_t *__restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;  // destination column x channel
  const int y = blockIdx.y;
  if (x >= W2 * 3) return;
  const int u = x / 3, c = x - 3 * u;
  const size_t sp = (size_t)W2 * 2 * 3;  // source pitch in bytes
  const uint8_t *r0 = src + (size_t)(2 * y) * sp + (size_t)(2 * u) * 3 + c;
  const uint8_t *r1 = r0 + sp;
  dst[(size_t)y * W2 * 3 + x] = (uint8_t)(((int)r0[0] + (int)r0[3] + (int)r1[0] + (int)r1[3] + 2) >> 2);
}

__global__ void ea_resize_half_f32_kernel(const float *__restrict__ src, int H2, int W2, float *__restrict__ dst,
                                          int nan_to_zero) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= W2) return;
  const size_t sp = (size_t)W2 * 2;
  const float *r0 = src + (size_t)(2 * y) * sp + 2 * x;
  const float *r1 = r0 + sp;
  float a = r0[0], b = r0[1], c = r1[0], d = r1[1];
  if (nan_to_zero) {  // depth.setTo(0, depth != depth) before the resize (src/ea.cpp:56-58)
    a = (a != a) ? 0.0f : a; b = (b != b) ? 0.0f : b; c = (c != c) ? 0.0f : c; d = (d != d) ? 0.0f : d;
  }
  dst[(size_t)y * W2 + x] = __fmul_rn(__fadd_rn(__fadd_rn(a, b), __fadd_rn(c, d)), 0.25f);
}

__global__ void ea_nan_to_zero_kernel(float *__restrict__ img, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = img[i]; if (v != v) img[i] = 0.0f; }
}

// ---- launchers ----------------------------------------------------------------------------------

hipError_t launch_resize_half_bgr8(const uint8_t *src, int H, int W, uint8_t *dst, hipStream_t s) {
  const int H2 = H / 2, W2 = W / 2;
  dim3 block(256), grid((W2 * 3 + 255) / 256, H2);
  hipLaunchKernelGGL(ea_resize_half_bgr8_kernel, grid, block, 0, s, src, H2, W2, dst);
  return hipGetLastError();
}

hipError_t launch_resize_half_f32(const float *src, int H, int W, float *dst, int nan_to_zero, hipStream_t s) {
  const int H2 = H / 2, W2 = W / 2;
  dim3 block(256), grid((W2 + 255) / 256, H2);
  hipLaunchKernelGGL(ea_resize_half_f32_kernel, grid, block, 0, s, src, H2, W2, dst, nan_to_zero);
  return hipGetLastError();
}

hipError_t launch_nan_to_zero(float *img, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(ea_nan_to_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, n);
  return hipGetLastError();
}


hipError_t launch_edge_strength(const uint8_t *bgr, int H, int W, uint8_t *gray, uint8_t *lap, hipStream_t s) {
  dim3 block(256), grid((W + 255) / 256, H);
  hipLaunchKernelGGL(ea_blur_gray_kernel, grid, block, 0, s, bgr, H, W, gray);
  hipLaunchKernelGGL(ea_laplacian_abs_kernel, grid, block, 0, s, gray, H, W, lap);
  return hipGetLastError();
}

hipError_t launch_threshold_median(const uint8_t *lap, int H, int W, int thr, int median, uint8_t *mask, hipStream_t s) {
  dim3 block(256), grid((W + 255) / 256, H);
  hipLaunchKernelGGL(ea_threshold_median_kernel, grid, block, 0, s, lap, H, W, thr, median, mask);
  return hipGetLastError();
}

// scratch: 4 * ceil(H/32) * W ints
// dist_fix (3x3 chamfer, 16.16 fixed point) or, when dist_f32 is given, the exact Euclidean distance in float32
hipError_t launch_chamfer(const uint8_t *mask, int H, int W, int *G, int *scratch, int *dist_fix, float *dist_f32,
                          unsigned int *minmax, hipStream_t s) {
  const int S = (H + kSegRows - 1) / kSegRows;
  int *end_dn = scratch, *end_up = scratch + (size_t)S * W, *carry_dn = scratch + 2 * (size_t)S * W,
      *carry_up = scratch + 3 * (size_t)S * W;
  hipLaunchKernelGGL(ea_column_local_kernel, dim3((W + 63) / 64, S), dim3(64), 0, s, mask, H, W, G, end_dn, end_up, minmax);
  hipLaunchKernelGGL(ea_column_carry_kernel, dim3((W + 63) / 64), dim3(64), 0, s, end_dn, end_up, H, W, S, carry_dn, carry_up);
  const int row_threads = std::min(1024, std::max(256, ((W + 63) / 64) * 64));
  if (dist_f32)
    hipLaunchKernelGGL(ea_edt_row_kernel, dim3(H), dim3(row_threads), (size_t)W * sizeof(int), s, G, carry_dn, carry_up, H, W,
                       dist_f32, minmax);
  else
    hipLaunchKernelGGL(ea_chamfer_row_kernel, dim3(H), dim3(row_threads), (size_t)W * sizeof(int), s, G, carry_dn, carry_up, H, W,
                       dist_fix, minmax);
  return hipGetLastError();
}

hipError_t launch_dt_store(int dtype, const int *dist_fix, const float *dist_f32, int H, int W, const unsigned int *minmax,
                           int normalize, double lo, double hi, void *dst, int pitch, float *plain, float *dst32, hipStream_t s) {
  dim3 block(256), grid((W + 2 * kImagePad + 255) / 256, H + 2 * kImagePad);
  if (dtype == 1)
    hipLaunchKernelGGL((ea_dt_store_kernel<float>), grid, block, 0, s, dist_fix, dist_f32, H, W, minmax, normalize, lo, hi, (float *)dst, pitch, plain, nullptr);
  else
    hipLaunchKernelGGL((ea_dt_store_kernel<double>), grid, block, 0, s, dist_fix, dist_f32, H, W, minmax, normalize, lo, hi, (double *)dst, pitch, plain, dst32);
  return hipGetLastError();
}

// blur 3x3 -> gray -> Canny(low, high): labels are left in `label` (2 = edge after the hysteresis), `edges` is the
// CV_8U edge map [AND keep > 1], `inv` = 255 - edges.  mag: H*W ints, dir/label/edges/inv: H*W bytes,
// changed: one int in device memory, h_changed: its pinned host mirror (polled between hysteresis rounds).
// l2_bgr != 0: the ROS flavour -- no blur / gray, Sobel on the three channels, L2 magnitude, (low, high) already squared
hipError_t launch_canny(const uint8_t *bgr, int H, int W, int low, int high, int l2_bgr, const uint8_t *keep, uint8_t *gray,
                        int *mag, uint8_t *dir, uint8_t *label, uint8_t *edges, uint8_t *inv, int *changed, int *rounds_out,
                        hipStream_t s) {
  dim3 block(256), grid((W + 255) / 256, H);
  if (l2_bgr) {
    hipLaunchKernelGGL(ea_sobel_mag_l2_bgr_kernel, grid, block, 0, s, bgr, H, W, mag, dir);
  } else {
    hipLaunchKernelGGL(ea_boxblur_gray_kernel, grid, block, 0, s, bgr, H, W, gray);
    hipLaunchKernelGGL(ea_sobel_mag_kernel, grid, block, 0, s, gray, H, W, mag, dir);
  }
  hipLaunchKernelGGL(ea_canny_nms_kernel, grid, block, 0, s, mag, dir, H, W, low, high, label, changed, kHystBatch);
  // hysteresis: launches are enqueued eight at a time, each with a change flag of its own; a launch whose predecessor
  // changed nothing returns at once, and the host reads the eight flags with one stream synchronisation
  const int tiles_x = (W + kHystTile - 1) / kHystTile, tiles = tiles_x * ((H + kHystTile - 1) / kHystTile);
  const dim3 tgrid((tiles + 3) / 4);
  int rounds = 0;
  for (;;) {
    hipError_t e = hipSuccess;
    if (rounds > 0) e = hipMemsetAsync(changed, 0, kHystBatch * sizeof(int), s);  // (the first batch's flags: zeroed by the NMS kernel)
    if (e != hipSuccess) return e;
    for (int k = 0; k < kHystBatch; ++k)
      hipLaunchKernelGGL(ea_canny_hysteresis_kernel, tgrid, dim3(256), 0, s, label, H, W, tiles_x, tiles,
                         k == 0 ? (const int *)nullptr : changed + k - 1, changed + k);
    int h[kHystBatch];
    e = hipMemcpyAsync(h, changed, kHystBatch * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    int quiet = -1;
    for (int k = 0; k < kHystBatch && quiet < 0; ++k)
      if (!h[k]) quiet = k;
    if (quiet >= 0) { rounds += quiet + 1; break; }  // launch `quiet` confirmed the fixed point
    rounds += kHystBatch;
    // safety net only: a launch that reports a change has grown the edge set, and a chain crosses at least one tile
    // per launch, so the number of tiles bounds the launches
    if (rounds > (int)(((long long)H * W) / 256) + 64) return hipErrorUnknown;
  }
  if (rounds_out) *rounds_out = rounds;
  const int npix = H * W;
  hipLaunchKernelGGL(ea_canny_finish_kernel, dim3((npix + 255) / 256), block, 0, s, label, keep, npix, edges, inv);
  return hipGetLastError();
}

hipError_t launch_edge_scatter_ros(int dtype, const uint8_t *edges, const float *depth, int H, int W, const int *block_offsets,
                                   double fx, double fy, double cx, double cy, void *X, void *Y, void *Z, int capacity,
                                   hipStream_t s) {
  const int npix = H * W, nblocks = (npix + kScanBlock - 1) / kScanBlock;
  if (dtype == 1)
    hipLaunchKernelGGL((ea_edge_scatter_ros_kernel<float>), dim3(nblocks), dim3(kScanBlock), 0, s, edges, depth, H, W,
                       block_offsets, fx, fy, cx, cy, (float *)X, (float *)Y, (float *)Z, capacity);
  else
    hipLaunchKernelGGL((ea_edge_scatter_ros_kernel<double>), dim3(nblocks), dim3(kScanBlock), 0, s, edges, depth, H, W,
                       block_offsets, fx, fy, cx, cy, (double *)X, (double *)Y, (double *)Z, capacity);
  return hipGetLastError();
}

hipError_t launch_gate_by_mask(uint8_t *grad, const uint8_t *mask, int H, int W, hipStream_t s) {
  const int npix = H * W;
  hipLaunchKernelGGL(ea_gate_by_mask_kernel, dim3((npix + 255) / 256), dim3(256), 0, s, grad, mask, npix);
  return hipGetLastError();
}

hipError_t launch_edge_count_scan(const uint8_t *lap, const uint16_t *depth, int H, int W, int thr, int *block_counts,
                                  int *total, hipStream_t s) {
  const int npix = H * W, nblocks = (npix + kScanBlock - 1) / kScanBlock;
  hipLaunchKernelGGL(ea_edge_count_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, lap, depth, npix, thr, block_counts);
  hipLaunchKernelGGL(ea_edge_scan_kernel, dim3(1), dim3(kScanBlock), 0, s, block_counts, nblocks, total);
  return hipGetLastError();
}

hipError_t launch_edge_scatter(int dtype, const uint8_t *lap, const uint16_t *depth, int H, int W, int thr,
                               const int *block_offsets, double fx, double fy, double cx, double cy, double z_scaling,
                               void *X, void *Y, void *Z, int capacity, hipStream_t s) {
  const int npix = H * W, nblocks = (npix + kScanBlock - 1) / kScanBlock;
  if (dtype == 1)
    hipLaunchKernelGGL((ea_edge_scatter_kernel<float>), dim3(nblocks), dim3(kScanBlock), 0, s, lap, depth, H, W, thr,
                       block_offsets, fx, fy, cx, cy, z_scaling, (float *)X, (float *)Y, (float *)Z, capacity);
  else
    hipLaunchKernelGGL((ea_edge_scatter_kernel<double>), dim3(nblocks), dim3(kScanBlock), 0, s, lap, depth, H, W, thr,
                       block_offsets, fx, fy, cx, cy, z_scaling, (double *)X, (double *)Y, (double *)Z, capacity);
  return hipGetLastError();
}

}  // namespace ea

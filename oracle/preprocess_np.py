"""numpy restatement of the reference's CPU pre-processing that PRODUCES the hot path's inputs.

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED except for one number: `get_aX` on the bundled frame 1
yields 44 457 edge points, i.e. ceil(44457/30) = 1482 residual blocks, the count in the
reference's only recorded log (standalone/README.md:34).  tests/test_golden_and_preprocess.py holds
that check.

Follows (ref = /root/reference):
  get_aX / get_aX_mask    ref: standalone/utils.cpp:201-281, :283-369
  get_distance_transform  ref: standalone/utils.cpp:38-83
  get_distance_transform2 / _masked / _NoNormalize / _masked_NoNormalize   ref: standalone/utils.cpp:85-199
  get_aX_canny            ref: standalone/utils.cpp:371-462
  SolveEA::setRefFrame / setNowFrame (ROS flavour)   ref: src/SolveEA.cpp:29-82, :86-119
and the published OpenCV 3 algorithms those call (OpenCV is absent from this image):
  blur 3x3 on 8-bit                  -> box sum / 9 rounded to nearest, reflect-101
  Canny(gray, 30, 90) (aperture 3, L1 gradient) -> Sobel 3x3 CV_16S with replicated border, |dx|+|dy|,
                                        non-maximum suppression with the tan(22.5 deg) fixed-point test
                                        (TG22 = 13573, shift 15), hysteresis over 8-neighbours
  Canny(bgr, 150, 100, 3, true)      -> per pixel the channel with the largest dx^2 + dy^2 (first on ties) supplies
                                        dx, dy; thresholds min(t, 32767)^2; the same suppression and hysteresis
  distanceTransform(DIST_L2, DIST_MASK_PRECISE) -> exact Euclidean: float32 sqrt(dx^2 + dy^2) of the nearest zero pixel
  GaussianBlur 3x3 sigma=0 on 8-bit  -> [1 2 1]x[1 2 1]/16, fixed point, round-half-up, reflect-101
  cvtColor CV_RGB2GRAY on 8-bit      -> (4899*c0 + 9617*c1 + 1868*c2 + 8192) >> 14  (c0 is really
                                        blue: imread returns BGR, the code says RGB — utils.cpp:51,216)
  Laplacian CV_16S ksize=3           -> [[2,0,2],[0,-8,0],[2,0,2]], reflect-101
  convertScaleAbs                    -> min(|x|, 255)
  medianBlur 3                       -> 3x3 median, replicate border
  distanceTransform(DIST_L2, 3)      -> two-pass 3x3 chamfer, a=0.955 b=1.3693 in 16.16 fixed point
  normalize(NORM_MINMAX, 0, 1)       -> float32  src*scale + shift
"""
import numpy as np


def load_rgb_as_bgr(path):
    """cv::imread(path) equivalent for an 8-bit colour PNG: HxWx3 uint8 in B,G,R order."""
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return im[:, :, ::-1].copy()


def load_depth_u16(path):
    """cv::imread(path, CV_LOAD_IMAGE_ANYDEPTH) for the TUM 16-bit depth PNG."""
    from PIL import Image
    im = Image.open(path)
    a = np.asarray(im)
    return a.astype(np.uint16)


def _pad_reflect101(a, n=1):
    return np.pad(a, ((n, n), (n, n)) + ((0, 0),) * (a.ndim - 2), mode="reflect")


def gaussian_blur3_u8(img):
    """cv::GaussianBlur(img, Size(3,3), 0, 0, BORDER_DEFAULT) on CV_8UC3 / CV_8UC1."""
    a = _pad_reflect101(img.astype(np.int32))
    h = a[:, :-2] + 2 * a[:, 1:-1] + a[:, 2:]
    v = h[:-2] + 2 * h[1:-1] + h[2:]
    return ((v + 8) >> 4).astype(np.uint8)


def rgb2gray_u8(img3):
    """cv::cvtColor(img, CV_RGB2GRAY) applied to whatever is in channels 0,1,2."""
    c = img3.astype(np.int64)
    return ((4899 * c[:, :, 0] + 9617 * c[:, :, 1] + 1868 * c[:, :, 2] + 8192) >> 14).astype(np.uint8)


def laplacian3_abs_u8(gray):
    """cv::Laplacian(gray, CV_16S, 3, 1, 0, BORDER_DEFAULT) then cv::convertScaleAbs."""
    a = _pad_reflect101(gray.astype(np.int32))
    lap = 2 * (a[:-2, :-2] + a[:-2, 2:] + a[2:, :-2] + a[2:, 2:]) - 8 * a[1:-1, 1:-1]
    return np.minimum(np.abs(lap), 255).astype(np.uint8)


def edge_strength(img_bgr):
    """The gradient map both get_aX and get_distance_transform threshold at > 35."""
    return laplacian3_abs_u8(rgb2gray_u8(gaussian_blur3_u8(img_bgr)))


def get_aX(img_bgr, depth_u16, fx, fy, cx, cy, z_scaling=5000.0, threshold=35, mask_u8=None):
    """ref: utils.cpp:201-281; with `mask_u8`: get_aX_mask, utils.cpp:283-369 (also requires mask > 0).
    Returns (a_X 4xN float64 in raster order, (v,u) index arrays)."""
    grad = edge_strength(img_bgr)
    H, W = grad.shape
    Z = depth_u16.astype(np.float64) / float(z_scaling)
    u = np.arange(W, dtype=np.float64)[None, :]
    v = np.arange(H, dtype=np.float64)[:, None]
    X = (u - cx) * Z / fx
    Y = (v - cy) * Z / fy
    keep = (grad.astype(np.float64) > threshold) & (Z > 0)
    if mask_u8 is not None:
        keep &= mask_u8 > 0
    vv, uu = np.nonzero(keep)  # raster order: v outer, u inner
    a_X = np.stack([X[vv, uu], Y[vv, uu], Z[vv, uu], np.ones(vv.size)], axis=0)
    return a_X, (vv, uu)


def median_blur3_u8(img):
    """cv::medianBlur(img, 3) on CV_8UC1 (replicate border)."""
    a = np.pad(img, 1, mode="edge")
    H, W = img.shape
    stack = np.stack([a[i:i + H, j:j + W] for i in range(3) for j in range(3)], axis=0)
    return np.sort(stack, axis=0)[4].astype(np.uint8)


HV_DIST = int(round(0.955 * 65536))     # CV_FLT_TO_FIX(0.955f, 16)
DIAG_DIST = int(round(1.3693 * 65536))  # CV_FLT_TO_FIX(1.3693f, 16)
_DIST_MAX = (1 << 31) // 2 - 1          # INT_MAX >> 1 scale guard


def _row_scan(c, step, reverse=False):
    """t[j] = min(c[j], t[j-1] + step) along the last axis (min-plus scan), vectorised."""
    if reverse:
        return _row_scan(c[..., ::-1], step)[..., ::-1]
    k = np.arange(c.shape[-1], dtype=np.int64) * step
    return np.minimum.accumulate(c - k, axis=-1) + k


def chamfer3x3_fixed(zero_mask):
    """cv::distanceTransform(src, DIST_L2, 3) core: src==0 pixels are sources.
    Returns int64 distances in 16.16 fixed point."""
    H, W = zero_mask.shape
    BIG = np.int64(_DIST_MAX)
    t = np.full((H + 2, W + 2), BIG, dtype=np.int64)
    # forward pass
    for i in range(1, H + 1):
        up = t[i - 1]
        c = np.minimum(np.minimum(up[:-2] + DIAG_DIST, up[1:-1] + HV_DIST), up[2:] + DIAG_DIST)
        c = np.where(zero_mask[i - 1], 0, c)
        # the left neighbour of column 1 is the border (BIG): the scan starts from c
        row = _row_scan(c, HV_DIST)
        row = np.where(zero_mask[i - 1], 0, row)
        t[i, 1:-1] = np.minimum(row, BIG)
    # backward pass
    for i in range(H, 0, -1):
        dn = t[i + 1]
        c = np.minimum(np.minimum(dn[2:] + DIAG_DIST, dn[1:-1] + HV_DIST), dn[:-2] + DIAG_DIST)
        c = np.minimum(c, t[i, 1:-1])
        row = _row_scan(c, HV_DIST, reverse=True)
        t[i, 1:-1] = np.minimum(row, BIG)
    return t[1:-1, 1:-1]


def distance_transform_l2_3(src_u8):
    d = chamfer3x3_fixed(src_u8 == 0)
    return (d.astype(np.float64) * (1.0 / 65536.0)).astype(np.float32)


def normalize_minmax_f32(dist, lo=0.0, hi=1.0):
    smin, smax = float(dist.min()), float(dist.max())
    scale = (hi - lo) * (1.0 / (smax - smin) if (smax - smin) > np.finfo(np.float64).eps else 0.0)
    shift = lo - smin * scale
    return (dist * np.float32(scale) + np.float32(shift)).astype(np.float32)


def get_distance_transform(img_bgr, threshold=35, normalize=True):
    """ref: utils.cpp:38-83.  Returns HxW float32 in [0,1]."""
    lap = edge_strength(img_bgr)
    B = np.where(lap > threshold, 0, 255).astype(np.uint8)
    Bf = median_blur3_u8(B)
    dist = distance_transform_l2_3(Bf)
    return normalize_minmax_f32(dist) if normalize else dist


def box_blur3_u8(img):
    """cv::blur(img, Size(3,3)) on CV_8UC3 / CV_8UC1: normalised box filter, BORDER_DEFAULT (reflect-101);
    the 8-bit result is the sum of nine divided by 9, rounded to nearest (9 is odd: no ties)."""
    a = _pad_reflect101(img.astype(np.int32))
    H, W = img.shape[:2]
    sm = sum(a[i:i + H, j:j + W] for i in range(3) for j in range(3))
    return ((sm + 4) // 9).astype(np.uint8)


def sobel3_s16_replicate(gray):
    """cv::Sobel(gray, CV_16S, 1, 0, 3) and (0, 1, 3) with BORDER_REPLICATE, as cv::Canny computes them."""
    a = np.pad(gray.astype(np.int32), 1, mode="edge")
    H, W = gray.shape
    p = lambda di, dj: a[1 + di:1 + di + H, 1 + dj:1 + dj + W]
    dx = (p(-1, 1) + 2 * p(0, 1) + p(1, 1)) - (p(-1, -1) + 2 * p(0, -1) + p(1, -1))
    dy = (p(1, -1) + 2 * p(1, 0) + p(1, 1)) - (p(-1, -1) + 2 * p(-1, 0) + p(-1, 1))
    return dx, dy


CANNY_SHIFT = 15
TG22 = int(0.4142135623730950488016887242097 * (1 << CANNY_SHIFT) + 0.5)  # 13573


def canny_nms_labels(img, low_thresh, high_thresh, l2_gradient=False):
    """Stages 1-2 of cv::Canny(img, edges, t1, t2, 3, l2_gradient) on CV_8UC1 or CV_8UC3: per-pixel label 2 = strong
    edge (local maximum above `high`), 0 = candidate (local maximum above `low`), 1 = not an edge.  Magnitudes outside
    the image are 0.  Multi-channel input: the channel with the largest magnitude (first on ties) supplies dx, dy."""
    lo_t, hi_t = sorted((float(low_thresh), float(high_thresh)))
    if l2_gradient:
        lo_t, hi_t = min(32767.0, lo_t), min(32767.0, hi_t)
        lo_t = lo_t * lo_t if lo_t > 0 else lo_t
        hi_t = hi_t * hi_t if hi_t > 0 else hi_t
    low, high = int(np.floor(lo_t)), int(np.floor(hi_t))
    chans = [img] if img.ndim == 2 else [img[:, :, c] for c in range(img.shape[2])]
    dx = dy = mag = None
    for ch in chans:
        cdx, cdy = sobel3_s16_replicate(ch)
        cm = (cdx * cdx + cdy * cdy) if l2_gradient else (np.abs(cdx) + np.abs(cdy))
        if mag is None:
            dx, dy, mag = cdx, cdy, cm
        else:
            better = cm > mag
            dx, dy, mag = np.where(better, cdx, dx), np.where(better, cdy, dy), np.where(better, cm, mag)
    H, W = mag.shape
    m = np.pad(mag, 1)  # zero border
    c = m[1:-1, 1:-1]
    left, right = m[1:-1, :-2], m[1:-1, 2:]
    up, down = m[:-2, 1:-1], m[2:, 1:-1]
    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << CANNY_SHIFT
    tg22x = x * TG22
    tg67x = tg22x + (x << (CANNY_SHIFT + 1))
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = ~(horiz | vert)
    s = np.where((dx ^ dy) < 0, -1, 1)
    # diagonal neighbours: row above at column j - s, row below at column j + s
    jj = np.arange(W)[None, :] + np.zeros((H, 1), dtype=np.int64)
    ii = np.arange(H)[:, None] + np.zeros((1, W), dtype=np.int64)
    up_d = m[ii, jj + 1 - s]       # m is padded by one: (i-1)+1 = ii, (j - s)+1
    down_d = m[ii + 2, jj + 1 + s]
    is_max = np.where(horiz, (c > left) & (c >= right), np.where(vert, (c > up) & (c >= down), (c > up_d) & (c > down_d)))
    is_max &= c > low
    labels = np.ones((H, W), dtype=np.uint8)
    labels[is_max] = 0
    labels[is_max & (c > high)] = 2
    return labels


def canny_hysteresis(labels):
    """Stage 3: candidates (0) 8-connected to a strong pixel (2) become edges.  Returns the CV_8U edge map."""
    from scipy import ndimage
    cand = labels != 1
    comp, n = ndimage.label(cand, structure=np.ones((3, 3), dtype=bool))
    keep = np.zeros(n + 1, dtype=bool)
    keep[np.unique(comp[labels == 2])] = True
    keep[0] = False
    return np.where(keep[comp], 255, 0).astype(np.uint8)


def canny_u8(img, low_thresh, high_thresh, l2_gradient=False):
    """cv::Canny(img, edges, low_thresh, high_thresh, 3, l2_gradient) on CV_8UC1 / CV_8UC3."""
    return canny_hysteresis(canny_nms_labels(img, low_thresh, high_thresh, l2_gradient))


def distance_transform_precise(src_u8):
    """cv::distanceTransform(src, DIST_L2, DIST_MASK_PRECISE): float32 sqrt of the exact squared Euclidean distance
    to the nearest zero pixel (the sum dx^2 + dy^2 formed in float32 like OpenCV's row pass; exact below 2^24).
    Needs at least one zero pixel."""
    from scipy import ndimage
    assert (src_u8 == 0).any(), "DIST_MASK_PRECISE on an image without zero pixels is not restated"
    _, (iy, ix) = ndimage.distance_transform_edt(src_u8 != 0, return_indices=True)
    H, W = src_u8.shape
    yy, xx = np.mgrid[0:H, 0:W]
    dx2 = ((xx - ix).astype(np.int64) ** 2).astype(np.float32)
    dy2 = ((yy - iy).astype(np.int64) ** 2).astype(np.float32)
    return np.sqrt(dx2 + dy2, dtype=np.float32)


def ros_ref_points(img_bgr, depth_f32, fx, fy, cx, cy, low=150.0, high=100.0):
    """SolveEA::setRefFrame (ref: src/SolveEA.cpp:29-82): Canny(rgb, 150, 100, 3, true) on the 3-channel image; every
    edge pixel gives a point, Z = depth (float, metres), Z == 0 -> 1.0.  Returns (3xN float64 raster order, (v, u))."""
    edges = canny_u8(img_bgr, low, high, l2_gradient=True)
    vv, uu = np.nonzero(edges)
    Z = depth_f32[vv, uu].astype(np.float64)
    Z = np.where(Z == 0, 1.0, Z)
    X = Z * (uu.astype(np.float64) - cx) / fx
    Y = Z * (vv.astype(np.float64) - cy) / fy
    return np.stack([X, Y, Z], axis=0), (vv, uu)


def ros_now_distance_transform(img_bgr, low=150.0, high=100.0):
    """SolveEA::setNowFrame (ref: src/SolveEA.cpp:86-119): Canny -> 255 - edges -> distanceTransform(L2, PRECISE) ->
    normalize to [0, 255]."""
    edges = canny_u8(img_bgr, low, high, l2_gradient=True)
    return normalize_minmax_f32(distance_transform_precise(255 - edges), 0.0, 255.0)


def canny_edges_of_frame(img_bgr, low=30, high=90):
    """blur 3x3 -> CV_RGB2GRAY -> Canny(30, 90): the edge map shared by get_distance_transform2* and get_aX_canny
    (ref: utils.cpp:87-94, :397-404)."""
    return canny_u8(rgb2gray_u8(box_blur3_u8(img_bgr)), low, high)


def get_distance_transform2(img_bgr, mask_u8=None, normalize=(0.0, 1.0)):
    """ref: utils.cpp:85-199.  `mask_u8` None: get_distance_transform2 (normalize (0,1), :85-106) or
    get_distance_transform2_NoNormalize (normalize None, :142-165); with a mask: the _masked variants
    (:108-141 normalises to (0,255), :166-199 does not): edges survive where inputmask > 1."""
    edges = canny_edges_of_frame(img_bgr)
    if mask_u8 is not None:
        edges = np.where(mask_u8 > 1, edges, 0).astype(np.uint8)
    dist = distance_transform_l2_3(255 - edges)
    return normalize_minmax_f32(dist, *normalize) if normalize is not None else dist


def get_aX_canny(img_bgr, depth_u16, fx, fy, cx, cy, z_scaling=5000.0):
    """ref: utils.cpp:371-462.  Returns (a_X 4xN float64 in raster order, (v,u) index arrays)."""
    edges = canny_edges_of_frame(img_bgr)
    H, W = edges.shape
    Z = depth_u16.astype(np.float64) / float(z_scaling)
    u = np.arange(W, dtype=np.float64)[None, :]
    v = np.arange(H, dtype=np.float64)[:, None]
    X = (u - cx) * Z / fx
    Y = (v - cy) * Z / fy
    keep = (edges > 0) & (Z > 0)
    vv, uu = np.nonzero(keep)
    a_X = np.stack([X[vv, uu], Y[vv, uu], Z[vv, uu], np.ones(vv.size)], axis=0)
    return a_X, (vv, uu)


def resize_half_bgr8(img_u8):
    """cv::resize(im, im_resized, cv::Size(), 0.5, 0.5) on a bgr8 frame (ref: src/ea.cpp:38).  At an exact factor of two
    OpenCV's INTER_LINEAR resize is its fast 2 x 2 area mean (cv::resize switches to INTER_AREA there; the fixed-point
    bilinear path gives the same numbers: weights 1024/2048 twice): (a + b + c + d + 2) >> 2 per channel."""
    a = img_u8.astype(np.int32)
    H, W = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
    a = a[:H, :W]
    s = a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]
    return ((s + 2) >> 2).astype(np.uint8)


def resize_half_f32(img_f32, nan_to_zero=True):
    """the depth callback (ref: src/ea.cpp:56-62): depth.setTo(0, depth != depth), then cv::resize x0.5 on CV_32F = the
    area mean in float32.  Summation order of OpenCV 3's vector path (ResizeAreaFastVec_SIMD_32f, what x86 / NEON builds
    run): the two pixels of each source row first, then the two rows, times 0.25f.  (The scalar generic loop would add
    ((a + b) + c) + d; without OpenCV in the image neither can be checked against the library itself.)"""
    a = np.array(img_f32, dtype=np.float32, copy=True)
    if nan_to_zero:
        a[np.isnan(a)] = np.float32(0)
    H, W = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
    a = a[:H, :W]
    top = (a[0::2, 0::2] + a[0::2, 1::2]).astype(np.float32)
    bot = (a[1::2, 0::2] + a[1::2, 1::2]).astype(np.float32)
    return ((top + bot).astype(np.float32) * np.float32(0.25)).astype(np.float32)


def grid_view_of_image(dt_hw):
    """What `ceres::Grid2D<double,1> grid(e.data(), 0, e.cols(), 0, e.rows())` sees after
    cv::cv2eigen into a column-major Eigen::MatrixXd (standalone_edge_align.cpp:205-206,258):
    a row-major array with num_rows = W (u), num_cols = H (v), value(r,c) = DT[v=c, u=r]."""
    return np.ascontiguousarray(dt_hw.astype(np.float64).T)

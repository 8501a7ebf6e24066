"""Vectorised numpy restatement of the hot path — a SECOND, independently derived checker.

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see oracle/ea_oracle.h).  It is deliberately written
from a different derivation than oracle/ea_oracle.c so the two can cross-check each other:
  * bicubic via explicit Catmull-Rom weight polynomials (not Horner on spline coefficients);
  * Jacobian via the unit-quaternion identity  d(b)/d(delta) = -2 [R a]x  (SURVEY App. A.4),
    valid for |q| = 1, whereas ea_oracle.c uses general dR/dq * P(q) and Jet autodiff.

Follows ref: standalone/utils.h:48-80 (functor), standalone_edge_align.cpp:258 (grid view).
"""
import numpy as np

LOSS_TRIVIAL, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _cr_weights(x):
    """Catmull-Rom weights of p0..p3 and their derivatives at fraction x."""
    x2, x3 = x * x, x * x * x
    w = np.stack([0.5 * (-x3 + 2 * x2 - x), 0.5 * (3 * x3 - 5 * x2 + 2),
                  0.5 * (-3 * x3 + 4 * x2 + x), 0.5 * (x3 - x2)], axis=0)
    d = np.stack([0.5 * (-3 * x2 + 4 * x - 1), 0.5 * (9 * x2 - 10 * x),
                  0.5 * (-9 * x2 + 8 * x + 1), 0.5 * (3 * x2 - 2 * x)], axis=0)
    return w, d


def bicubic(grid, r, c):
    """Vectorised ceres::BiCubicInterpolator<Grid2D<double,1>>::Evaluate on a row-major grid."""
    grid = np.asarray(grid, dtype=np.float64)
    rows, cols = grid.shape
    r = np.asarray(r, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    row = np.floor(np.clip(r, -1e9, 1e9)).astype(np.int64)
    col = np.floor(np.clip(c, -1e9, 1e9)).astype(np.int64)
    wr, dr = _cr_weights(r - row)
    wc, dc = _cr_weights(c - col)
    f = np.zeros_like(r)
    dfdr = np.zeros_like(r)
    dfdc = np.zeros_like(r)
    for k in range(4):
        ri = np.clip(row - 1 + k, 0, rows - 1)
        for l in range(4):
            ci = np.clip(col - 1 + l, 0, cols - 1)
            p = grid[ri, ci]
            f += wr[k] * wc[l] * p
            dfdr += dr[k] * wc[l] * p
            dfdc += wr[k] * dc[l] * p
    return f, dfdr, dfdc


def loss(kind, a, s):
    s = np.asarray(s, dtype=np.float64)
    if kind == LOSS_CAUCHY:
        b = a * a
        return b * np.log1p(s / b), 1.0 / (1.0 + s / b)
    if kind == LOSS_HUBER:
        b = a * a
        rt = np.sqrt(np.maximum(s, 1e-300))
        out = s > b
        return np.where(out, 2 * a * rt - b, s), np.where(out, a / rt, 1.0)
    return s, np.ones_like(s)


def evaluate(grid, K, xyz, q, t, loss_kind=LOSS_CAUCHY, loss_a=1.0, z_guard=0.01):
    """Returns dict(raw_r, raw_J(n,6), r, J, cost, JtJ, Jtr, valid).  q must be unit."""
    fx, fy, cx, cy = K
    X = np.asarray(xyz, dtype=np.float64)[:, :3]
    R = quat_to_R(np.asarray(q, dtype=np.float64))
    t = np.asarray(t, dtype=np.float64)
    cvec = X @ R.T
    b = cvec + t
    valid = ~((b[:, 2] < z_guard) & (b[:, 2] > -z_guard)) if z_guard > 0 else np.ones(len(X), bool)
    bz = np.where(valid, b[:, 2], 1.0)
    u = fx * b[:, 0] / bz + cx
    v = fy * b[:, 1] / bz + cy
    f, Fu, Fv = bicubic(grid, u, v)
    g = np.stack([Fu * fx / bz, Fv * fy / bz,
                  -(Fu * fx * b[:, 0] + Fv * fy * b[:, 1]) / (bz * bz)], axis=1)
    J = np.concatenate([2.0 * np.cross(cvec, g), g], axis=1)
    rho, w = loss(loss_kind, loss_a, f * f)
    sq = np.sqrt(w)
    rc = sq * f
    Jc = J * sq[:, None]
    m = valid
    return dict(raw_r=f, raw_J=J, r=rc, J=Jc, valid=valid,
                cost=0.5 * float(np.sum(rho[m])),
                JtJ=Jc[m].T @ Jc[m], Jtr=Jc[m].T @ rc[m], n_invalid=int((~m).sum()))


def pixel_cost(xyz, q, t, fx, fy, cx, cy, dt_image):
    """The reference's integer-pixel cost report (ref: standalone_edge_align.cpp:2494-2567, :2704-2776): location3D =
    b_T_a * a_X; unvn = (x / z, y / z, 1); imagePixel = K * unvn; cost = disTrans.at<float>((int)v, (int)u); total, mean,
    max and the pixel of the first maximum.  Points outside the image (which upstream reads unchecked) are skipped and
    counted.  dt_image: H x W, [v][u].  Returns dict(total_cost, mean_cost, max_cost, max_pixel, count, outside)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    R = quat_to_R(np.asarray(q, dtype=np.float64))
    t = np.asarray(t, dtype=np.float64)
    b = np.empty_like(xyz[:, :3])
    for k in range(3):  # row by row, left to right, as Eigen's product evaluates it
        b[:, k] = ((R[k, 0] * xyz[:, 0] + R[k, 1] * xyz[:, 1]) + R[k, 2] * xyz[:, 2]) + t[k]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = fx * (b[:, 0] / b[:, 2]) + cx
        v = fy * (b[:, 1] / b[:, 2]) + cy
    H, W = dt_image.shape
    fin = np.isfinite(u) & np.isfinite(v) & (np.abs(u) < 1e9) & (np.abs(v) < 1e9)
    iu = np.where(fin, np.trunc(np.where(fin, u, 0)), -1).astype(np.int64)
    iv = np.where(fin, np.trunc(np.where(fin, v, 0)), -1).astype(np.int64)
    ok = fin & (iu >= 0) & (iu < W) & (iv >= 0) & (iv < H) & (u > -1.0) & (v > -1.0)
    cost = np.asarray(dt_image, dtype=np.float64)[iv[ok], iu[ok]]
    out = dict(total_cost=float(cost.sum()), count=int(ok.sum()), outside=int((~ok).sum()), max_cost=-1.0, max_pixel=(0.0, 0.0))
    out["mean_cost"] = out["total_cost"] / out["count"] if out["count"] else 0.0
    if out["count"]:
        k = int(np.argmax(cost))  # first maximum
        idx = np.flatnonzero(ok)[k]
        out["max_cost"] = float(cost[k]); out["max_pixel"] = (float(u[idx]), float(v[idx]))
    return out

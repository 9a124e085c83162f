"""Second, independent restatement of the round-3 primitives -- test infrastructure (round 4).

Same rules as cv2411_numpy.py: written from the published OpenCV 2.4.11 algorithms (imgproc/src/color.cpp
`YUV420sp2RGB888Invoker`, imgproc/src/undistort.cpp `initUndistortRectifyMap`, imgproc/src/imgwarp.cpp `remap` with
CV_16SC2 + CV_16UC1 maps) and, for the lens composition, from the specification in `oracle/mst_oracle.h` -- WITHOUT
consulting `oracle/undistort.c`.  `tests/test_independent_round3.py` diff-tests them against the oracle.
"""
import numpy as np

from . import cv2411_numpy as N

F = np.float32


def cvt_nv12_bgr(yuv):
    """cvtColor(yuv, COLOR_YUV2BGR_NV12): yuv = (h * 3 / 2, w) uint8 (h rows of Y, h / 2 rows of interleaved U V).
    ITU-R BT.601 in 20-bit fixed point; U V are shared by a 2 x 2 block of luma samples."""
    yuv = np.asarray(yuv, np.uint8)
    w = yuv.shape[1]
    h = yuv.shape[0] * 2 // 3
    CY, CUB, CUG, CVG, CVR, SH = 1220542, 2116026, -409993, -852492, 1673527, 20
    y = np.maximum(yuv[:h].astype(np.int64) - 16, 0) * CY
    uv = yuv[h:h + h // 2].astype(np.int64)
    u = np.repeat(np.repeat(uv[:, 0::2] - 128, 2, axis=0), 2, axis=1)
    v = np.repeat(np.repeat(uv[:, 1::2] - 128, 2, axis=0), 2, axis=1)
    half = 1 << (SH - 1)
    r = (y + half + CVR * v) >> SH
    g = (y + half + CVG * v + CUG * u) >> SH
    b = (y + half + CUB * u) >> SH
    return np.stack([N.saturate_u8(b), N.saturate_u8(g), N.saturate_u8(r)], axis=2)


def _inv3(m):
    """cv::invert of a 3 x 3 CV_64F matrix (DECOMP_LU takes the closed form for n <= 3): cofactors times 1 / det."""
    m = np.asarray(m, np.float64).reshape(3, 3)
    d = m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) + \
        m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0])
    d = 1.0 / d
    t = np.empty((3, 3), np.float64)
    t[0, 0] = (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d
    t[0, 1] = (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d
    t[0, 2] = (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d
    t[1, 0] = (m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d
    t[1, 1] = (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d
    t[1, 2] = (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d
    t[2, 0] = (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d
    t[2, 1] = (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d
    t[2, 2] = (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d
    return t


def init_undistort_rectify_map(K, dist, size, R=None, newK=None):
    """initUndistortRectifyMap(K, dist, R, newK, size, CV_16SC2): (map1 (h, w, 2) int16, map2 (h, w) uint16).
    The homogeneous coordinate is ACCUMULATED along a row (_x += ir[0] per column), all in double; the fixed-point
    split is iu = cvRound(u * 32): map1 = iu >> 5 cast to short, map2 = (iv & 31) * 32 + (iu & 31)."""
    w, h = size
    K = np.asarray(K, np.float64).reshape(3, 3)
    A = K if newK is None else np.asarray(newK, np.float64).reshape(3, 3)
    Rm = np.eye(3) if R is None else np.asarray(R, np.float64).reshape(3, 3)
    d = list(np.asarray(dist, np.float64).reshape(-1)) + [0.0] * 8
    k1, k2, p1, p2, k3, k4, k5, k6 = d[:8]
    # (A * R) as cv::Mat's operator* forms it: plain double dot products
    AR = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            AR[i, j] = A[i, 0] * Rm[0, j] + A[i, 1] * Rm[1, j] + A[i, 2] * Rm[2, j]
    ir = _inv3(AR).reshape(9)
    fx, fy, u0, v0 = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    m1 = np.empty((h, w, 2), np.int16)
    m2 = np.empty((h, w), np.uint16)
    for i in range(h):
        x0, y0, w0 = i * ir[1] + ir[2], i * ir[4] + ir[5], i * ir[7] + ir[8]
        # the accumulation _x += ir[0]: a running double sum, reproduced with cumulative sums seeded by the row start
        xs = np.cumsum(np.concatenate(([x0], np.full(w - 1, ir[0]))))
        ys = np.cumsum(np.concatenate(([y0], np.full(w - 1, ir[3]))))
        ws = np.cumsum(np.concatenate(([w0], np.full(w - 1, ir[6]))))
        iw = 1.0 / ws
        x, y = xs * iw, ys * iw
        x2, y2 = x * x, y * y
        r2, _2xy = x2 + y2, 2 * x * y
        kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
        u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0
        v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0
        iu, iv = N.cv_round(u * 32), N.cv_round(v * 32)
        m1[i, :, 0] = (iu >> 5).astype(np.int64).astype(np.int16)          # plain (short) cast: wraps
        m1[i, :, 1] = (iv >> 5).astype(np.int64).astype(np.int16)
        m2[i] = ((iv & 31) * 32 + (iu & 31)).astype(np.uint16)
    return m1, m2


def remap_fixed(src, map1, map2):
    """remap(src, dst, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR, BORDER_CONSTANT, Scalar()) for CV_8U images."""
    src = np.asarray(src, np.uint8)
    s3 = src if src.ndim == 3 else src[:, :, None]
    sh, sw, cn = s3.shape
    sx, sy = map1[..., 0].astype(np.int64), map1[..., 1].astype(np.int64)
    fxy = map2.astype(np.int64) & 1023
    tab = N._bilinear_tab()
    wgt = tab[fxy >> 5, fxy & 31]
    acc = np.zeros(sx.shape + (cn,), np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        ix, iy = sx + dx, sy + dy
        ok = (ix >= 0) & (ix < sw) & (iy >= 0) & (iy < sh)
        v = np.where(ok[..., None], s3[np.clip(iy, 0, sh - 1), np.clip(ix, 0, sw - 1)].astype(np.int64), 0)
        acc += v * wgt[..., k:k + 1]
    out = N.saturate_u8((acc + (1 << 14)) >> 15)
    return out if src.ndim == 3 else out[:, :, 0]


def compose_lens_maps(xmap, ymap, map1, map2):
    """The fused lens variant's composition as oracle/mst_oracle.h specifies it (this project's own definition)."""
    h, w = map2.shape
    lx = (map1[..., 0].astype(F) + (map2 & 31).astype(F) / F(32)).astype(F)
    ly = (map1[..., 1].astype(F) + ((map2 >> 5) & 31).astype(F) / F(32)).astype(F)
    u = np.clip(np.asarray(xmap, F), F(0), F(w - 1)).astype(F)
    v = np.clip(np.asarray(ymap, F), F(0), F(h - 1)).astype(F)
    u = np.where(np.isnan(np.asarray(xmap, F)), F(0), u)
    v = np.where(np.isnan(np.asarray(ymap, F)), F(0), v)
    x0 = np.minimum(u.astype(np.int64), max(w - 2, 0))
    y0 = np.minimum(v.astype(np.int64), max(h - 2, 0))
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    ax, ay = (u - x0.astype(F)).astype(F), (v - y0.astype(F)).astype(F)

    def lerp2(m):
        r0 = ((m[y0, x0] * (F(1) - ax)).astype(F) + (m[y0, x1] * ax).astype(F)).astype(F)
        r1 = ((m[y1, x0] * (F(1) - ax)).astype(F) + (m[y1, x1] * ax).astype(F)).astype(F)
        return ((r0 * (F(1) - ay)).astype(F) + (r1 * ay).astype(F)).astype(F)
    return lerp2(lx), lerp2(ly)

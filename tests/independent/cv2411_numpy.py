"""A SECOND, independent restatement of the OpenCV 2.4.11 primitives the per-frame path rests on -- test infrastructure.

Why it exists: the reference links prebuilt OpenCV 2.4.11 binaries; neither their source nor any OpenCV is in this
image, so `oracle/*.c` restates `remap`, `pyrDown`, `pyrUp`, `resize`, `copyMakeBorder`, `distanceTransform`,
`dilate` ... from the published algorithm, and parity is "unpinned" (DESIGN.md section 2).  This module is a second
reading of the same published algorithms, written in numpy in a separate pass WITHOUT consulting `oracle/*.c` (sources:
the 2.4.11 `imgproc/src/{imgwarp,pyramids,distransform,morph}.cpp` and `core/src/{copy,convert}.cpp` as published,
and SURVEY.md section 8c's notes).  `tests/test_independent_restatement.py` diff-tests the two on random inputs.  Two
readings that agree bit for bit do not lift "unpinned" -- both could share a misreading of the original -- but they
remove the single-author risk for everything an implementation slip would have changed.

Conventions: images are numpy arrays (H, W) or (H, W, C); every function returns a new array.
"""
import numpy as np

BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
INTER_NEAREST, INTER_LINEAR = 0, 1


# ---- scalar helpers -----------------------------------------------------------------------------------------
def cv_round(a):
    """cvRound on x64 (cvtsd2si): round half to even, 0x80000000 when the value does not fit an int."""
    a = np.asarray(a, np.float64)
    r = np.rint(a)
    bad = ~((r >= -2147483648.0) & (r <= 2147483647.0))
    return np.where(bad, -2147483648, np.where(bad, 0, r)).astype(np.int64)


def saturate_u8(v):
    return np.clip(v, 0, 255).astype(np.uint8)


def saturate_s16(v):
    return np.clip(v, -32768, 32767).astype(np.int16)


def border_interpolate(p, n, border):
    """cv::borderInterpolate for integer arrays; -1 marks a BORDER_CONSTANT miss."""
    p = np.array(p, np.int64, copy=True)
    if border == BORDER_CONSTANT:
        return np.where((p >= 0) & (p < n), p, -1)
    if border == BORDER_REPLICATE:
        return np.clip(p, 0, n - 1)
    if border == BORDER_WRAP:
        return np.mod(p, n)
    if n == 1:
        return np.zeros_like(p)
    delta = 1 if border == BORDER_REFLECT_101 else 0
    while True:
        lo, hi = p < 0, p >= n
        if not (lo.any() or hi.any()):
            return p
        p = np.where(lo, -p - 1 + delta, np.where(hi, n - 1 - (p - n) - delta, p))


# ---- cv::remap, CV_8U, float maps --------------------------------------------------------------------------------
def _bilinear_tab():
    """BilinearTab_i of initInterTab2D(INTER_LINEAR, fixpt): 32x32 entries of four shorts, each the float product of
    the 1-D weights times 2^15 saturated to short, then nudged so that every entry sums to 2^15 -- which only touches
    the (0, 0) entry: {32768 -> 32767, 0, 0, 0} has its deficit added to the largest / smallest member of the block
    scan, leaving {32767, 0, 0, 1}."""
    tab = np.zeros((32, 32, 4), np.int64)
    for fy in range(32):
        for fx in range(32):
            wy, wx = (1.0 - fy / 32.0, fy / 32.0), (1.0 - fx / 32.0, fx / 32.0)
            w = [np.float32(np.float32(wy[k1]) * np.float32(wx[k2])) for k1 in range(2) for k2 in range(2)]
            it = [int(min(max(int(np.rint(float(v) * 32768.0)), -32768), 32767)) for v in w]
            tab[fy, fx] = it
    # the only entry whose shorts do not sum to 32768 is (0, 0): 32767 + 0 + 0 + 0.  The fix-up scan of the reference
    # walks k1, k2 in [ksize/2, ksize/2 + 2) over the flat table, i.e. elements 3, 4, 5, 6 relative to the entry (its
    # own last weight and the first three of the NEXT entry, which is still zero-filled at that point), and adds the
    # deficit to the largest of them: element 3, as nothing later is larger than its zero
    assert tab[0, 0].sum() == 32767
    tab[0, 0, 3] += 1
    assert (tab.sum(axis=2) == 32768).all()
    return tab


_TAB = None


def remap_8u(src, xmap, ymap, interp, border):
    """cv::remap(src, dst, xmap, ymap, interp, border, Scalar()) for CV_8UC1 / CV_8UC3 and CV_32FC1 maps."""
    global _TAB
    src = np.asarray(src, np.uint8)
    s3 = src if src.ndim == 3 else src[:, :, None]
    sh, sw, cn = s3.shape
    xmap = np.asarray(xmap, np.float32)
    ymap = np.asarray(ymap, np.float32)
    if interp == INTER_NEAREST:
        sx = np.clip(cv_round(xmap), -32768, 32767)
        sy = np.clip(cv_round(ymap), -32768, 32767)
        ix, iy = border_interpolate(sx, sw, border), border_interpolate(sy, sh, border)
        ok = (ix >= 0) & (iy >= 0)
        out = np.where(ok[..., None], s3[np.clip(iy, 0, sh - 1), np.clip(ix, 0, sw - 1)], 0).astype(np.uint8)
        return out if src.ndim == 3 else out[:, :, 0]
    if _TAB is None:
        _TAB = _bilinear_tab()
    # cvRound(map * INTER_TAB_SIZE): a float times an int constant, formed in float
    fsx = cv_round((xmap * np.float32(32.0)).astype(np.float64))
    fsy = cv_round((ymap * np.float32(32.0)).astype(np.float64))
    sx = np.clip(fsx >> 5, -32768, 32767)
    sy = np.clip(fsy >> 5, -32768, 32767)
    w = _TAB[fsy & 31, fsx & 31]                       # (H, W, 4)
    acc = np.zeros(xmap.shape + (cn,), np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        ix, iy = border_interpolate(sx + dx, sw, border), border_interpolate(sy + dy, sh, border)
        ok = (ix >= 0) & (iy >= 0)
        v = np.where(ok[..., None], s3[np.clip(iy, 0, sh - 1), np.clip(ix, 0, sw - 1)].astype(np.int64), 0)
        acc += v * w[..., k:k + 1]
    if border == BORDER_CONSTANT:
        # a window wholly outside the image is the constant itself (same value: all four taps are 0 anyway)
        pass
    out = saturate_u8((acc + (1 << 14)) >> 15)
    return out if src.ndim == 3 else out[:, :, 0]


# ---- cv::copyMakeBorder ----------------------------------------------------------------------------------------
def copy_make_border(src, top, bottom, left, right, border, value=0):
    src = np.asarray(src)
    h, w = src.shape[:2]
    iy = border_interpolate(np.arange(-top, h + bottom), h, border)
    ix = border_interpolate(np.arange(-left, w + right), w, border)
    out = src[np.clip(iy, 0, h - 1)][:, np.clip(ix, 0, w - 1)].copy()
    if border == BORDER_CONSTANT:
        out[iy < 0] = value
        out[:, ix < 0] = value
    return out


# ---- cv::pyrDown / cv::pyrUp --------------------------------------------------------------------------------------
def pyr_down_int(src):
    """pyrDown of CV_8U / CV_16S: [1 4 6 4 1] x [1 4 6 4 1] in int, BORDER_REFLECT_101, (v + 128) >> 8."""
    src = np.asarray(src)
    a = src.astype(np.int64)
    h, w = a.shape[:2]
    dh, dw = (h + 1) // 2, (w + 1) // 2
    cols = [border_interpolate(2 * np.arange(dw) + k, w, BORDER_REFLECT_101) for k in (-2, -1, 0, 1, 2)]
    rows = [border_interpolate(2 * np.arange(dh) + k, h, BORDER_REFLECT_101) for k in (-2, -1, 0, 1, 2)]
    hz = a[:, cols[2]] * 6 + (a[:, cols[1]] + a[:, cols[3]]) * 4 + a[:, cols[0]] + a[:, cols[4]]
    v = hz[rows[2]] * 6 + (hz[rows[1]] + hz[rows[3]]) * 4 + hz[rows[0]] + hz[rows[4]]
    out = (v + 128) >> 8
    return saturate_u8(out) if src.dtype == np.uint8 else saturate_s16(out)


def pyr_down_32f(src):
    """pyrDown of CV_32FC1 as the x64 SSE build evaluates it: the horizontal pass is scalar
    (s0*6 + (s-1 + s+1)*4 + s-2 + s+2, left to right); the vertical pass runs PyrDownVec_32f on the first
    width - width % 8 columns -- (r0 + r4 + (r2 + r2)) + (r1 + r3 + r2) * 4 -- and the scalar expression
    r2*6 + (r1 + r3)*4 + r0 + r4 on the rest; both then multiply by 1/256."""
    a = np.asarray(src, np.float32)
    h, w = a.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    f = np.float32
    cols = [border_interpolate(2 * np.arange(dw) + k, w, BORDER_REFLECT_101) for k in (-2, -1, 0, 1, 2)]
    rows = [border_interpolate(2 * np.arange(dh) + k, h, BORDER_REFLECT_101) for k in (-2, -1, 0, 1, 2)]
    s = [a[:, c] for c in cols]
    hz = ((s[2] * f(6) + (s[1] + s[3]) * f(4)) + s[0]) + s[4]
    r = [hz[rr] for rr in rows]
    nvec = dw - dw % 8
    out = np.empty((dh, dw), np.float32)
    t = (r[0] + r[4]) + (r[2] + r[2])
    u = (r[1] + r[3]) + r[2]
    vec = (t + u * f(4)) * f(1.0 / 256.0)
    sca = (((r[2] * f(6) + (r[1] + r[3]) * f(4)) + r[0]) + r[4]) * f(1.0 / 256.0)
    out[:, :nvec] = vec[:, :nvec]
    out[:, nvec:] = sca[:, nvec:]
    return out


def pyr_up_16s(src):
    """pyrUp of CV_16S to exactly twice the size: even outputs s[-1] + 6 s[0] + s[+1], odd ones 4 (s[0] + s[+1]) in
    each direction, (v + 32) >> 6, saturated; left edge 6 s0 + 2 s1, right edge s[n-2] + 7 s[n-1] and 8 s[n-1];
    rows index through reflect-101 of 2y on the destination grid."""
    a = np.asarray(src, np.int16).astype(np.int64)
    h, w = a.shape[:2]
    row = np.zeros((h, 2 * w) + a.shape[2:], np.int64)
    if w == 1:
        row[:, 0] = a[:, 0] * 8
        row[:, 1] = a[:, 0] * 8
    else:
        row[:, 0] = a[:, 0] * 6 + a[:, 1] * 2
        row[:, 1] = (a[:, 0] + a[:, 1]) * 4
        row[:, 2:2 * w - 2:2] = a[:, :-2] + a[:, 1:-1] * 6 + a[:, 2:]
        row[:, 3:2 * w - 2:2] = (a[:, 1:-1] + a[:, 2:]) * 4
        row[:, 2 * w - 2] = a[:, w - 2] + a[:, w - 1] * 7
        row[:, 2 * w - 1] = a[:, w - 1] * 8
    idx = lambda sy: border_interpolate(np.asarray(sy) * 2, 2 * h, BORDER_REFLECT_101) // 2
    i = np.arange(h)
    r0, r1, r2 = row[idx(i - 1)], row[idx(i)], row[idx(i + 1)]
    out = np.zeros((2 * h,) + row.shape[1:], np.int64)
    out[0::2] = (r1 * 6 + r0 + r2 + 32) >> 6
    out[1::2] = ((r1 + r2) * 4 + 32) >> 6
    return saturate_s16(out)


# ---- cv::resize(INTER_LINEAR) to an explicit size ------------------------------------------------------------------
def _linear_axis(src_len, dst_len, clamp_fraction):
    """Index / weight tables of resizeGeneric_ for one axis (explicit dsize: inv_scale = dst/src as double, scale its
    reciprocal).  The x axis zeroes the fraction where it clamps the index (clamp_fraction); the y axis only clamps."""
    inv = float(dst_len) / float(src_len)
    scale = 1.0 / inv
    ofs = np.zeros(dst_len, np.int64)
    frac = np.zeros(dst_len, np.float32)
    for d in range(dst_len):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if clamp_fraction:
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= src_len - 1:
                s, f = src_len - 1, np.float32(0)
        ofs[d], frac[d] = s, f
    return ofs, frac


def resize_linear_32f(src, dw, dh):
    a = np.asarray(src, np.float32)
    h, w = a.shape
    f = np.float32
    xo, xf = _linear_axis(w, dw, True)
    yo, yf = _linear_axis(h, dh, False)
    a0, a1 = f(1) - xf, xf
    x1 = np.minimum(xo + 1, w - 1)
    hz = a[:, xo] * a0[None, :] + a[:, x1] * a1[None, :]
    r0 = hz[np.clip(yo, 0, h - 1)]
    r1 = hz[np.clip(yo + 1, 0, h - 1)]
    b0, b1 = (f(1) - yf)[:, None], yf[:, None]
    return r0 * b0 + r1 * b1


def resize_linear_8u(src, dw, dh):
    """cv::resize(INTER_LINEAR) of CV_8UC1 to an explicit size: 11-bit fixed-point weights (saturate_cast<short> of
    weight * 2048), integer horizontal pass, and the vertical pass of VResizeLinear<uchar, int, short>:
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    a = np.asarray(src, np.uint8).astype(np.int64)
    h, w = a.shape
    xo, xf = _linear_axis(w, dw, True)
    yo, yf = _linear_axis(h, dh, False)
    q = lambda v: np.clip(np.rint(v.astype(np.float64) * 2048.0), -32768, 32767).astype(np.int64)
    a0, a1 = q(np.float32(1) - xf), q(xf)
    b0, b1 = q(np.float32(1) - yf), q(yf)
    x1 = np.minimum(xo + 1, w - 1)
    hz = a[:, xo] * a0[None, :] + a[:, x1] * a1[None, :]
    s0 = hz[np.clip(yo, 0, h - 1)]
    s1 = hz[np.clip(yo + 1, 0, h - 1)]
    out = (((b0[:, None] * (s0 >> 4)) >> 16) + ((b1[:, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return saturate_u8(out)


# ---- exposure --------------------------------------------------------------------------------------------------
def blocks_gain_apply(img, gain_map):
    """BlocksGainCompensator::apply (exposure_compensate.cpp:225-246): the gain map resized to the image, then per
    channel saturate_cast<uchar>(pixel * gain) = clamp(cvRound(float product))."""
    img = np.asarray(img, np.uint8)
    g = resize_linear_32f(gain_map, img.shape[1], img.shape[0])
    prod = img.astype(np.float32) * g[:, :, None]
    r = cv_round(prod.astype(np.float64))
    return np.where(np.isnan(prod), 0, np.clip(r, 0, 255)).astype(np.uint8)


# ---- cv::distanceTransform(CV_DIST_L1, 3) and createWeightMap --------------------------------------------------------
def distance_transform_l1(mask):
    """The 3x3 chamfer of distanceTransform for CV_DIST_L1: integer distances scaled by 2^16 (1 for an edge step, 2 for
    a diagonal), a one-pixel frame of "infinity" (INT_MAX >> 2), a forward and a backward raster pass, result * 2^-16."""
    m = np.asarray(mask)
    h, w = m.shape
    INF, A, B = (2 ** 31 - 1) >> 2, 1 << 16, 2 << 16
    t = np.full((h + 2, w + 2), INF, np.int64)
    for y in range(1, h + 1):
        for x in range(1, w + 1):
            if m[y - 1, x - 1] == 0:
                t[y, x] = 0
            else:
                t[y, x] = min(t[y - 1, x - 1] + B, t[y - 1, x] + A, t[y - 1, x + 1] + B, t[y, x - 1] + A)
    out = np.zeros((h, w), np.float32)
    for y in range(h, 0, -1):
        for x in range(w, 0, -1):
            v = t[y, x]
            if v > A:
                v = min(v, t[y + 1, x + 1] + B, t[y + 1, x] + A, t[y + 1, x - 1] + B, t[y, x + 1] + A)
                t[y, x] = v
            out[y - 1, x - 1] = np.float32(np.float32(v) * np.float32(1.0 / 65536.0))
    return out


def create_weight_map(mask, sharpness):
    """blenders.cpp:427-432: distanceTransform, then threshold(weight * sharpness, 1, THRESH_TRUNC)."""
    d = distance_transform_l1(mask)
    s = d * np.float32(sharpness) + np.float32(0)
    return np.where(s > np.float32(1), np.float32(1), s).astype(np.float32)


# ---- cv::dilate(src, dst, Mat()) -------------------------------------------------------------------------------------
def dilate3x3(src):
    """3x3 rectangular structuring element, anchor at the centre, border pixels ignored (the default border value of
    morphology is the identity of max)."""
    a = np.asarray(src)
    p = np.pad(a, 1, mode="constant", constant_values=np.iinfo(a.dtype).min if a.dtype.kind in "iu" else -np.inf)
    out = a.copy()
    h, w = a.shape
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, p[dy:dy + h, dx:dx + w])
    return out


# ---- saturating arithmetic on CV_16S -----------------------------------------------------------------------------------
def add_16s(a, b):
    return saturate_s16(np.asarray(a, np.int64) + np.asarray(b, np.int64))


def subtract_16s(a, b):
    return saturate_s16(np.asarray(a, np.int64) - np.asarray(b, np.int64))

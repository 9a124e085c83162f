"""Known-answer tests that pin the CPU oracle's OpenCV-2.4.11 primitive semantics (SURVEY.md 8c notes).

Every expected value below is hand-derived from the published 2.4.11 algorithm (fixed-point remap,
[1 4 6 4 1] pyramids with (v+128)>>8 / (v+32)>>6, saturating 16S arithmetic, round-half-even
conversions) or from SURVEY.md Appendix B / BASELINE.md section 3 (independently derived geometry).
The reference itself ships no tests or golden vectors: parity is "unpinned" beyond these.
"""
import numpy as np
import pytest

import oracle_lib as O

L = O.lib()


def test_cv_round_is_half_even_and_indefinite_on_overflow():
    assert [L.mo_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 1e3 + 0.5)] == [0, 2, 2, 0, -2, 2, 1000]
    assert L.mo_cv_round(3e9) == -2 ** 31 and L.mo_cv_round(-3e9) == -2 ** 31
    assert L.mo_cv_round(float("nan")) == -2 ** 31


def test_trunc_s16_is_cvttss2si_low_word():
    assert [L.mo_trunc_s16(v) for v in (1.9, -1.9, 0.999, -0.999, 32767.9)] == [1, -1, 0, 0, 32767]
    assert L.mo_trunc_s16(40000.0) == 40000 - 65536          # wraps, does not saturate
    assert L.mo_trunc_s16(-40000.0) == -40000 + 65536
    assert L.mo_trunc_s16(3e9) == 0 and L.mo_trunc_s16(float("inf")) == 0     # 0x80000000 -> low word 0


def test_border_interpolate():
    n = 10
    assert [L.mo_border_interpolate(p, n, O.BORDER_REFLECT) for p in (-1, -2, n, n + 1, 3)] == [0, 1, n - 1, n - 2, 3]
    assert [L.mo_border_interpolate(p, n, O.BORDER_REFLECT_101) for p in (-1, -2, n, n + 1)] == [1, 2, n - 2, n - 3]
    assert L.mo_border_interpolate(-1, n, O.BORDER_CONSTANT) == -1
    assert L.mo_border_interpolate(-32768, 1920, O.BORDER_REFLECT) in range(1920)
    assert L.mo_border_interpolate(-5, 1, O.BORDER_REFLECT) == 0


def test_bilinear_table():
    tab = np.ctypeslib.as_array(L.mo_bilinear_tab_i(), (1024, 4))
    assert (tab.sum(1) == 32768).all()
    assert tab[0].tolist() == [32767, 0, 0, 1]                 # OpenCV's sum fix-up quirk at fx=fy=0
    assert tab[16 * 32 + 16].tolist() == [8192] * 4
    fy, fx = np.divmod(np.arange(1, 1024), 32)
    exp = np.stack([(32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy], 1) * 32
    assert np.array_equal(tab[1:], exp)


def test_bilinear_quirk_is_unobservable_on_u8():
    v0, v3 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    assert np.array_equal((32767 * v0 + v3 + 16384) >> 15, v0)


def _maps(h, w, dx=0.0, dy=0.0):
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    return xs + np.float32(dx), ys + np.float32(dy)


def test_remap_identity_half_pixel_and_borders():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    xm, ym = _maps(7, 9)
    assert np.array_equal(O.remap(img, xm, ym, O.INTER_LINEAR, O.BORDER_REFLECT), img)
    xm, ym = _maps(7, 9, dx=0.5)
    out = O.remap(img, xm, ym, O.INTER_LINEAR, O.BORDER_REFLECT).astype(int)
    a = img.astype(int)
    right = np.concatenate([a[:, 1:], a[:, -1:]], 1)           # REFLECT: column 9 -> 8
    assert np.array_equal(out, (a + right + 1) >> 1)
    out = O.remap(img, xm, ym, O.INTER_LINEAR, O.BORDER_CONSTANT).astype(int)
    right0 = np.concatenate([a[:, 1:], np.zeros_like(a[:, -1:])], 1)
    assert np.array_equal(out, (a + right0 + 1) >> 1)
    # (-1,-1) -- what mapBackward returns for z <= 0 -- samples pixel (0,0) under BORDER_REFLECT
    m1 = np.full((2, 2), -1, np.float32)
    assert (O.remap(img, m1, m1, O.INTER_LINEAR, O.BORDER_REFLECT) == img[0, 0]).all()
    assert (O.remap(img, m1, m1, O.INTER_LINEAR, O.BORDER_CONSTANT) == 0).all()
    # 1/32-pixel quantisation: 3 + 1/64 rounds (half-even) to 3, 3 + 3/64 to 3 + 2/32
    g = np.arange(0, 160, 16, dtype=np.uint8)[None, :].repeat(3, 0)
    xm = np.array([[3 + 1 / 64, 3 + 3 / 64]], np.float32)
    ym = np.ones((1, 2), np.float32)
    assert O.remap(g, xm, ym, O.INTER_LINEAR, O.BORDER_REFLECT).tolist() == [[48, 49]]


def test_remap_nearest_rounds_half_even_and_constant_border():
    g = np.arange(10, dtype=np.uint8)[None, :] + np.zeros((3, 1), np.uint8) + 100
    xm = np.array([[0.5, 1.5, 2.5, -0.5, -0.51, 9.49, 9.5]], np.float32)
    ym = np.zeros((1, 7), np.float32)
    assert O.remap(g, xm, ym, O.INTER_NEAREST, O.BORDER_CONSTANT).tolist() == [[100, 102, 102, 100, 0, 109, 0]]


def test_pyr_down_16s():
    c = np.full((8, 12, 3), -777, np.int16)
    assert (O.pyr_down_16s(c) == -777).all()
    imp = np.zeros((9, 9), np.int16)
    imp[4, 4] = 256
    k = np.array([1, 4, 6, 4, 1])
    out = O.pyr_down_16s(imp)
    assert out.shape == (5, 5)
    exp = np.zeros((5, 5), int)
    for y in range(5):
        for x in range(5):
            ky, kx = 4 - 2 * y + 2, 4 - 2 * x + 2
            if 0 <= ky < 5 and 0 <= kx < 5:
                exp[y, x] = (k[ky] * k[kx] * 256 + 128) >> 8
    assert np.array_equal(out, exp)
    ramp = (4 * np.arange(16))[None, :].repeat(6, 0).astype(np.int16)
    assert np.array_equal(O.pyr_down_16s(ramp)[:, 1:-1], (8 * np.arange(8))[None, 1:-1].repeat(3, 0))
    # REFLECT_101 at the left edge: taps -2,-1 -> 2,1
    row = np.array([[10, 20, 30, 40]], np.int16)
    v = (6 * 10 + 4 * (20 + 20) + 30 + 30)          # horizontal
    assert O.pyr_down_16s(row)[0, 0] == (v * 16 + 128) >> 8     # single row: all 5 vertical taps hit it
    neg = np.full((4, 4), -1, np.int16)
    assert (O.pyr_down_16s(neg) == -1).all()        # arithmetic shift: (-256+128)>>8 == -1


def test_pyr_up_16s():
    c = np.full((3, 5, 3), 1234, np.int16)
    assert (O.pyr_up_16s(c) == 1234).all()
    imp = np.zeros((5, 5), np.int16)
    imp[2, 2] = 64
    up = O.pyr_up_16s(imp).astype(int)
    assert up[4, 4] == 36 and up[5, 4] == 24 and up[4, 5] == 24 and up[5, 5] == 16 and up[3, 3] == 16
    assert up[4, 3] == 24 and up[2, 4] == 6 and up[2, 2] == 1 and up[6, 6] == 1 and up[7, 7] == 0
    assert up.sum() == 64 * 4
    row = np.array([[64, 128, 0, -64]], np.int16).repeat(3, 0)
    u = O.pyr_up_16s(row)[2].astype(int)             # middle rows: vertical pass is the identity on constants
    s = [64, 128, 0, -64]
    exp = [6 * s[0] + 2 * s[1], 4 * (s[0] + s[1]), s[0] + 6 * s[1] + s[2], 4 * (s[1] + s[2]),
           s[1] + 6 * s[2] + s[3], 4 * (s[2] + s[3]), s[2] + 7 * s[3], 8 * s[3]]
    assert u.tolist() == [(8 * e + 32) >> 6 for e in exp]
    one = np.array([[7]], np.int16)
    assert (O.pyr_up_16s(one) == 7).all()


def test_pyr_down_32f_constant_and_sse_tail_split():
    ones = np.ones((16, 22), np.float32)            # dst width 11: 8 "SSE" columns + 3 scalar-tail columns
    assert (O.pyr_down_32f(ones) == 1.0).all()
    rng = np.random.default_rng(5)
    a = rng.random((16, 22), dtype=np.float32)
    out = O.pyr_down_32f(a)

    def refl(p, n):
        return -p if p < 0 else (2 * n - 2 - p if p >= n else p)
    f = np.float32
    for (y, x) in [(3, 2), (3, 9), (0, 0), (7, 10)]:
        rows = []
        for ky in range(-2, 3):
            r = a[refl(2 * y + ky, 16)]
            t = [r[refl(2 * x + kx, 22)] for kx in range(-2, 3)]
            rows.append(f(f(f(t[2] * f(6)) + f(f(t[1] + t[3]) * f(4))) + t[0]) + t[4])
        if x < 8:
            v = f(f(f(rows[0] + rows[4]) + f(rows[2] + rows[2])) + f(f(f(rows[1] + rows[3]) + rows[2]) * f(4)))
        else:
            v = f(f(f(f(rows[2] * f(6)) + f(f(rows[1] + rows[3]) * f(4))) + rows[0]) + rows[4])
        assert out[y, x] == f(v * f(1 / 256))


def test_saturating_add_subtract():
    a = np.array([32767, -32768, 100, -5], np.int16)
    b = np.array([1, 1, -200, 5], np.int16)
    out = np.empty(4, np.int16)
    L.mo_add_16s(a.ctypes.data, b.ctypes.data, out.ctypes.data, 4)
    assert out.tolist() == [32767, -32767, -100, 0]
    L.mo_subtract_16s(a.ctypes.data, b.ctypes.data, out.ctypes.data, 4)
    assert out.tolist() == [32766, -32768, 300, -10]


def test_resize_linear_32f():
    src = np.array([[0.0, 1.0]], np.float32)
    assert O.resize_linear_32f(src, 4, 1).tolist() == [[0.0, 0.25, 0.75, 1.0]]
    src = np.array([[0.0], [2.0]], np.float32)
    assert O.resize_linear_32f(src, 1, 4)[:, 0].tolist() == [0.0, 0.5, 1.5, 2.0]
    g = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(O.resize_linear_32f(g, 4, 3), g)


def test_gain_apply_rounds_half_even_and_saturates():
    img = np.array([[[100, 101, 200]]], np.uint8)
    assert O.blocks_gain_apply(img, np.array([[1.5]], np.float32)).tolist() == [[[150, 152, 255]]]   # 151.5 -> 152
    img = np.array([[[1, 3, 5]]], np.uint8)
    assert O.blocks_gain_apply(img, np.array([[0.5]], np.float32)).tolist() == [[[0, 2, 2]]]          # .5 -> even
    assert O.gain_apply(img, 2.5).tolist() == [[[2, 8, 12]]]                                          # 2.5->2, 7.5->8, 12.5->12


def test_distance_transform_l1_and_weight_map():
    m = np.full((5, 7), 255, np.uint8)
    m[2, 3] = 0
    d = O.distance_transform_l1(m)
    ys, xs = np.mgrid[0:5, 0:7]
    assert np.array_equal(d, (abs(ys - 2) + abs(xs - 3)).astype(np.float32))
    w = O.create_weight_map(m, 0.25)
    assert np.array_equal(w, np.minimum(d * np.float32(0.25), 1).astype(np.float32))
    full = np.full((4, 4), 255, np.uint8)
    assert (O.distance_transform_l1(full) > 8000).all()         # no zero pixel: INT_MAX>>2 seed survives
    assert (O.create_weight_map(full, 0.02) == 1).all()


def test_geometry_matches_survey_appendix_b():
    from stitchingvideo_amd import rig as R
    K = R.intrinsics(1250.0, 1920, 1080)
    rois = [O.warp_roi(O.WARP_SPHERICAL, 1250.0, K, R.rot_y(y), 1920, 1080) for y in (0, 60, 120, 180, -120, -60)]
    assert rois[0] == (-818, 1453, 1637, 1020) and rois[1] == (490, 1453, 1638, 1020)
    assert rois[3] == (-3925, 1453, 7852, 1020) and rois[4] == (-3436, 1453, 1638, 1020)
    pad = (-3925, 1453, 7872, 1024)
    assert O.multiband_feed_geometry(pad, 5, rois[0][:2], rois[0][2:]) == [-917, 1453, 939, 2477, 0, 99, 4, 120]
    assert O.multiband_feed_geometry(pad, 5, rois[1][:2], rois[1][2:])[4:] == [0, 127, 4, 123]
    assert O.multiband_feed_geometry(pad, 5, rois[3][:2], rois[3][2:])[4:] == [0, 0, 4, 20]
    assert O.multiband_feed_geometry(pad, 5, rois[4][:2], rois[4][2:])[4:] == [0, 105, 4, 113]
    # BASELINE.md section 3 expected canvases
    p = [O.warp_roi(O.WARP_PLANE, 1250.0, K, R.rot_y(y), 1920, 1080) for y in (-15, 15)]
    assert [r[2:] for r in p] == [(2148, 1406), (2147, 1406)]
    K4 = R.intrinsics(2500.0, 3840, 2160)
    assert O.warp_roi(O.WARP_CYLINDRICAL, 2500.0, K4, R.rot_y(180), 3840, 2160)[2:] == (15706, 2159)
    K8 = R.intrinsics(1800.0, 1920, 1080)
    assert O.warp_roi(O.WARP_SPHERICAL, 1800.0, K8, R.rot_y(180), 1920, 1080)[2:] == (11308, 1050)


def test_blender_no_paste_and_mask_or():
    b = O.Blender(O.BLEND_NO)
    b.prepare([(0, 0), (2, 0)], [(4, 2), (4, 2)])
    a = np.full((2, 4, 3), 10, np.int16)
    c = np.full((2, 4, 3), 20, np.int16)
    ma = np.array([[255, 255, 255, 0]] * 2, np.uint8)
    mc = np.array([[0, 7, 255, 255]] * 2, np.uint8)
    b.feed(a, ma, (0, 0))
    b.feed(c, mc, (2, 0))
    rect, dst, mask = b.blend()
    assert rect == (0, 0, 6, 2)
    assert dst[0, :, 0].tolist() == [10, 10, 10, 20, 20, 20] and mask[0].tolist() == [255, 255, 255, 7, 255, 255]


def test_feather_and_multiband_accumulate_semantics():
    # feather: two fully-overlapping constant images with all-255 masks -> weights 1 each -> mean, truncated
    b = O.Blender(O.BLEND_FEATHER, sharpness=1.0)
    b.prepare([(0, 0), (0, 0)], [(6, 6), (6, 6)])
    m = np.full((6, 6), 255, np.uint8)
    b.feed(np.full((6, 6, 3), 101, np.int16), m, (0, 0))
    b.feed(np.full((6, 6, 3), 200, np.int16), m, (0, 0))
    _, dst, mask = b.blend()
    assert (mask == 255).all() and (dst == int(301 / (2 + 1e-5))).all()      # 150 (trunc of 150.499)
    # multi-band on a constant image reproduces the constant inside the mask
    b = O.Blender(O.BLEND_MULTI_BAND, num_bands=3)
    b.prepare([(0, 0)], [(40, 24)])
    assert b.num_bands() == 3 and b.dst_roi() == (0, 0, 40, 24)
    b.feed(np.full((24, 40, 3), 77, np.int16), np.full((24, 40), 255, np.uint8), (0, 0))
    _, dst, mask = b.blend()
    assert (mask == 255).all() and (abs(dst.astype(int) - 77) <= 1).all()
    # band count is cropped to ceil(log2(max side)) (blenders.cpp:207-209)
    b = O.Blender(O.BLEND_MULTI_BAND, num_bands=9)
    b.prepare([(0, 0)], [(20, 10)])
    assert b.num_bands() == 5


def test_exposure_feed_known_answers():
    """GainCompensator::feed on hand-checkable inputs: identical overlapping images -> equal gains; cv::solve's LU
    against numpy on a well-conditioned system; the two-image closed form."""
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 4, 9):
        A = rng.uniform(-1, 1, size=(n, n)) + np.eye(n) * 4
        b = rng.uniform(-1, 1, size=n)
        ok, x = O.solve_lu(A, b)
        assert ok and np.allclose(A @ x, b, atol=1e-12)
    ok, x = O.solve_lu(np.zeros((4, 4)), np.ones(4))
    assert not ok and not x.any()
    # two images with the same content over a shared overlap: I(0,1) == I(1,0) -> gains equal (and ~1: beta pulls to 1)
    img = rng.integers(30, 200, size=(40, 60, 3), dtype=np.uint8)
    m = np.full((40, 60), 255, np.uint8)
    g = O.gain_feed([(0, 0), (20, 0)], [img, np.roll(img, -20, axis=1)], [m, m])
    assert abs(g[0] - g[1]) < 1e-9 and abs(g[0] - 1.0) < 1e-6
    # brighter second image -> its gain below the first one's
    bright = np.clip(np.roll(img, -20, axis=1).astype(int) * 3 // 2, 0, 255).astype(np.uint8)
    g = O.gain_feed([(0, 0), (20, 0)], [img, bright], [m, m])
    assert g[1] < g[0]
    # gain maps: constant gains stay constant under the [1/4 1/2 1/4] smoothing (weights sum to one exactly)
    maps = O.blocks_gain_feed([(0, 0)], [img], [m], block=(16, 16))
    assert maps[0].shape == (3, 4) and np.all(maps[0] == maps[0][0, 0])


def test_voronoi_seams_two_restatements_agree():
    """The oracle's C VoronoiSeamFinder against rig.py's scipy-based one (written independently for the rig builder)."""
    from stitchingvideo_amd import rig as rigmod
    for name, div in (("C3", 8), ("C2", 4), ("C4", 10)):
        cfg = rigmod.scaled_config(name, div)
        open_rig, cut_rig = rigmod.build_rig(cfg, seams=False), rigmod.build_rig(cfg, seams=True)
        got = O.voronoi_seams(open_rig["corners"], [c["mask"] for c in open_rig["cameras"]])
        assert all(np.array_equal(a, c["mask"]) for a, c in zip(got, cut_rig["cameras"]))


def test_dp_seam_finder_properties_and_two_rectangle_case():
    """DpSeamFinder restatement (oracle/dpseam.c).  No reference-held vector exists for it (parity unpinned); what is
    pinned here: it resolves every overlap (no pixel kept by two cameras, none dropped that a camera covered), it is
    deterministic, its two cost functions differ, and the hand-traced case of two side-by-side rectangles: the seam
    between the intersection and the FIRST-only component starts and ends where the two image contours cross (the
    intersection's left corners), hugs that side, and the intersection then goes to the second image as a whole
    (states INTERS -> INTERS_SECOND, seam_finders.cpp:443)."""
    from stitchingvideo_amd import rig as rigmod
    import helpers as H
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8), seams=False)
    frames = rigmod.frames_for(rig, 0)
    imgs, masks = [], []
    for c, f in zip(rig["cameras"], frames):
        _, w = O.warp(H.WARP_O[rig["warp"]], rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        imgs.append(w.astype(np.float32))
        masks.append(c["mask"].copy())
    a = O.dp_seams(rig["corners"], imgs, masks, "color")
    b = O.dp_seams(rig["corners"], imgs, masks, "color")
    g = O.dp_seams(rig["corners"], imgs, masks, "color_grad")
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert any(not np.array_equal(x, y) for x, y in zip(a, g))
    cx, cy = min(c[0] for c in rig["corners"]), min(c[1] for c in rig["corners"])
    W = max(c[0] + m.shape[1] for c, m in zip(rig["corners"], a)) - cx
    Hh = max(c[1] + m.shape[0] for c, m in zip(rig["corners"], a)) - cy
    for out in (a, g):
        cover, cover0 = np.zeros((Hh, W), np.int32), np.zeros((Hh, W), np.int32)
        for (x, y), m, m0 in zip(rig["corners"], out, masks):
            cover[y - cy:y - cy + m.shape[0], x - cx:x - cx + m.shape[1]] += (m != 0)
            cover0[y - cy:y - cy + m.shape[0], x - cx:x - cx + m.shape[1]] += (m0 != 0)
        assert cover0.max() == 2 and cover.max() == 1
        assert not ((cover == 0) & (cover0 > 0)).any()
    # two rectangles, 12 columns of overlap
    h, w, ov = 20, 30, 12
    im1, im2 = np.full((h, w, 3), 100, np.float32), np.full((h, w, 3), 200, np.float32)
    m1, m2 = np.full((h, w), 255, np.uint8), np.full((h, w), 255, np.uint8)
    o1, o2 = O.dp_seams([(0, 0), (w - ov, 0)], [im1, im2], [m1, m2], "color")
    assert (o1[:, :w - ov] == 255).all() and (o1[:, w - ov:] == 0).all() and (o2 == 255).all()

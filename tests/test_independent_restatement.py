"""Diff test of the two restatements of the OpenCV 2.4.11 primitives: oracle/*.c (the parity oracle) against
tests/independent/cv2411_numpy.py (a second reading of the published algorithms, written without the oracle's
sources).  Random inputs, bit for bit.  Agreement does not pin either to the real OpenCV 2.4.11 binaries (nothing
in this image can), it removes the single-author risk; any disagreement found while writing this was resolved by
going back to the published algorithm and is recorded in DESIGN.md section 2."""
import numpy as np
import pytest

import oracle_lib as O
from independent import cv2411_numpy as N

RNG = np.random.default_rng(20261003)


def _maps(h, w, sw, sh, rng, wild=False):
    xs = rng.uniform(-3, sw + 3, size=(h, w)).astype(np.float32)
    ys = rng.uniform(-3, sh + 3, size=(h, w)).astype(np.float32)
    # exact grid points, exact half-way ties of the 1/32 grid, and the warpers' (-1, -1) "behind the camera" marker
    xs[0, :8] = np.arange(8, dtype=np.float32)
    ys[0, :8] = 2.0
    xs[1, :8] = np.arange(8, dtype=np.float32) + np.float32(1.0 / 64.0)
    ys[1, :8] = np.float32(3.0 + 3.0 / 64.0)
    xs[2, :4], ys[2, :4] = -1.0, -1.0
    if wild:
        xs[3, :6] = [1e9, -1e9, np.nan, 40000.0, -40000.0, np.inf]
        ys[3, :6] = [1.0, 1.0, 1.0, 2.0, 2.0, 2.0]
    return xs, ys


@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("interp,border", [(N.INTER_LINEAR, N.BORDER_REFLECT), (N.INTER_LINEAR, N.BORDER_CONSTANT),
                                           (N.INTER_NEAREST, N.BORDER_CONSTANT), (N.INTER_NEAREST, N.BORDER_REFLECT)])
def test_remap(cn, interp, border):
    for sw, sh in ((37, 29), (5, 4), (1, 1), (2, 7)):
        src = RNG.integers(0, 256, size=(sh, sw) + ((cn,) if cn == 3 else ()), dtype=np.uint8)
        xs, ys = _maps(24, 31, sw, sh, RNG, wild=True)
        a = O.remap(src, xs, ys, interp, border)
        b = N.remap_8u(src, xs, ys, interp, border)
        assert np.array_equal(a, b), (sw, sh, np.argwhere(a != b)[:5])


def test_bilinear_table_entry_zero():
    t = N._bilinear_tab()
    assert list(t[0, 0]) == [32767, 0, 0, 1]
    assert list(t[16, 16]) == [8192] * 4 and list(t[0, 31]) == [1024, 31744, 0, 0]


@pytest.mark.parametrize("shape", [(16, 16), (17, 23), (5, 64), (33, 9), (2, 2), (1, 7), (3, 1)])
def test_pyr_down_16s_and_8u_ranges(shape):
    for cn in (1, 3):
        a = RNG.integers(-32768, 32768, size=shape + ((cn,) if cn == 3 else ())).astype(np.int16)
        assert np.array_equal(O.pyr_down_16s(a), N.pyr_down_int(a))
        b = RNG.integers(-255, 256, size=shape + ((cn,) if cn == 3 else ())).astype(np.int16)
        assert np.array_equal(O.pyr_down_16s(b), N.pyr_down_int(b))


@pytest.mark.parametrize("shape", [(16, 16), (18, 40), (7, 33), (64, 70), (3, 17), (2, 2)])
def test_pyr_down_32f(shape):
    a = RNG.uniform(0, 1, size=shape).astype(np.float32)
    x, y = O.pyr_down_32f(a), N.pyr_down_32f(a)
    assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    m = (RNG.integers(0, 2, size=shape) * 255).astype(np.float32) * np.float32(1.0 / 255.0)
    assert np.array_equal(O.pyr_down_32f(m).view(np.uint32), N.pyr_down_32f(m).view(np.uint32))


@pytest.mark.parametrize("shape", [(8, 8), (5, 9), (1, 4), (6, 1), (2, 2), (16, 3)])
def test_pyr_up_16s(shape):
    for cn in (1, 3):
        a = RNG.integers(-32768, 32768, size=shape + ((cn,) if cn == 3 else ())).astype(np.int16)
        assert np.array_equal(O.pyr_up_16s(a), N.pyr_up_16s(a))
        b = RNG.integers(-300, 300, size=shape + ((cn,) if cn == 3 else ())).astype(np.int16)
        assert np.array_equal(O.pyr_up_16s(b), N.pyr_up_16s(b))


@pytest.mark.parametrize("border", [N.BORDER_REFLECT, N.BORDER_CONSTANT, N.BORDER_REFLECT_101, N.BORDER_REPLICATE])
def test_copy_make_border(border):
    for shape, pads in (((6, 9), (3, 4, 5, 2)), ((4, 4), (9, 7, 11, 13)), ((1, 5), (2, 2, 3, 3)), ((7, 1), (0, 3, 2, 0))):
        a = RNG.integers(-500, 500, size=shape + (3,)).astype(np.int16)
        assert np.array_equal(O.copy_make_border_16s(a, *pads, border), N.copy_make_border(a, *pads, border))


@pytest.mark.parametrize("src,dst", [((32, 52), (1020, 1638)), ((5, 7), (64, 61)), ((3, 3), (3, 3)), ((9, 4), (5, 2)),
                                     ((1, 6), (20, 20)), ((6, 1), (13, 17)), ((40, 40), (31, 77))])
def test_resize_linear_32f(src, dst):
    a = RNG.uniform(0.5, 1.5, size=src).astype(np.float32)
    x, y = O.resize_linear_32f(a, dst[1], dst[0]), N.resize_linear_32f(a, dst[1], dst[0])
    assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_blocks_gain_apply():
    img = RNG.integers(0, 256, size=(70, 93, 3), dtype=np.uint8)
    for g in (RNG.uniform(0.7, 1.4, size=(3, 3)), RNG.uniform(0.0, 3.0, size=(9, 12)), np.full((2, 2), 0.5)):
        g = g.astype(np.float32)
        assert np.array_equal(O.blocks_gain_apply(img, g), N.blocks_gain_apply(img, g))
    tie = np.full((4, 4, 3), 1, np.uint8) * np.array([1, 3, 5], np.uint8)          # x.5 products: half to even
    assert np.array_equal(O.blocks_gain_apply(tie, np.full((2, 2), 0.5, np.float32)),
                          N.blocks_gain_apply(tie, np.full((2, 2), 0.5, np.float32)))


def test_distance_transform_and_weight_map():
    for shape, p in (((23, 31), 0.9), ((40, 17), 0.5), ((9, 9), 1.0), ((12, 12), 0.0)):
        m = (RNG.uniform(size=shape) < p).astype(np.uint8) * 255
        x, y = O.distance_transform_l1(m), N.distance_transform_l1(m)
        if not m.all():                                   # (no zero pixel: both hold "infinity", compared below too)
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        else:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        for sharp in (0.02, 0.3):
            a, b = O.create_weight_map(m, sharp), N.create_weight_map(m, sharp)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_dilate_and_resize_8u_mask_fixup():
    """stitcher.cpp:291-294: dilate(seam mask) -> resize to the warped size -> & warped mask."""
    for (h, w), (dh, dw) in (((20, 31), (163, 255)), ((7, 5), (64, 40)), ((16, 16), (16, 16)), ((3, 9), (40, 37))):
        m = (RNG.uniform(size=(h, w)) < 0.6).astype(np.uint8) * 255
        assert np.array_equal(O.dilate3x3(m), N.dilate3x3(m))
        g = RNG.integers(0, 256, size=(h, w), dtype=np.uint8)
        assert np.array_equal(O.dilate3x3(g), N.dilate3x3(g))
        assert np.array_equal(O.resize_linear_8u(g, dw, dh), N.resize_linear_8u(g, dw, dh))
        assert np.array_equal(O.resize_linear_8u(m, dw, dh), N.resize_linear_8u(m, dw, dh))

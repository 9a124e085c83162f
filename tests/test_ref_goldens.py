"""The oracle -- and on a GPU the HIP engine -- against golden vectors computed BY THE REFERENCE ITSELF: tests/golden/ref_runs.npz holds
inputs made here and the outputs the reference's shipped x64 machine code returned for them when run in place (tests/ref_run.py,
tests/make_ref_goldens.py; generated in the container that has /root/reference).  This file needs no reference tree: it runs on the
GPU box too."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_runs.npz")


@pytest.fixture(scope="module")
def g():
    with np.load(G) as z:
        return {k: z[k] for k in z.files}


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_oracle_remap_is_the_references(g):
    for name, border in (("reflect", 2), ("replicate", 1), ("reflect101", 4)):
        assert np.array_equal(O.remap(g["remap_src"], g["remap_xmap"], g["remap_ymap"], 1, border), g["remap_out_" + name]), name
    assert np.array_equal(O.remap_fixed(g["remap_src"], g["remapc_xy"], g["remapc_fxy"]), g["remapc_out"])            # BORDER_CONSTANT, CV_16SC2 maps
    assert np.array_equal(O.remap(g["remapg_src"], g["remapg_xmap"], g["remapg_ymap"], 1, 2), g["remapg_out_reflect"])  # one channel
    # INTER_NEAREST: an image and the all-255 mask of stitcher.cpp:278-280 (BORDER_CONSTANT), and BORDER_REFLECT
    assert np.array_equal(O.remap(g["remap_src"], g["remapn_xmap"], g["remapn_ymap"], 0, 0), g["remapn_out_constant"])
    assert np.array_equal(O.remap(np.full((61, 83), 255, np.uint8), g["remapn_xmap"], g["remapn_ymap"], 0, 0), g["remapn_mask_out_constant"])
    assert np.array_equal(O.remap(g["remap_src"], g["remapn_xmap"], g["remapn_ymap"], 0, 2), g["remapn_out_reflect"])
    roi, wx, wy = O.build_maps(1, 150.0, g["warp_K"], g["warp_R"], 160, 90)
    assert tuple(roi) == tuple(g["warp_roi"])
    assert np.array_equal(O.remap(g["warp_frame"], wx, wy, 1, 2), g["warp_out"])
    assert np.array_equal(O.remap(np.full((90, 160), 255, np.uint8), wx, wy, 0, 0), g["warp_mask_out"])


def test_oracle_pyramids_are_the_references(g):
    for i in range(5):
        assert np.array_equal(O.pyr_down_16s(g["pd16_src_%d" % i]), g["pd16_out_%d" % i]), i
    for i in range(4):
        assert np.array_equal(O.pyr_up_16s(g["pu16_src_%d" % i]), g["pu16_out_%d" % i]), i
        assert np.array_equal(_bits(O.pyr_down_32f(g["pd32_src_%d" % i])), _bits(g["pd32_out_%d" % i])), i


def test_oracle_normalise_and_conversions_are_the_references(g):
    v = g["norm_src"].copy()
    w = np.ascontiguousarray(g["norm_weight"])
    O.lib().mo_normalize_using_weight_map_32f(w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), w.shape[1], w.shape[0])
    assert np.array_equal(v, g["norm_out"])
    v, w = g["norm16_src"].copy(), np.ascontiguousarray(g["norm16_weight"])           # the CV_16S weight form: (v << 8) / (w + 1)
    O.lib().mo_normalize_using_weight_map_16s.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    O.lib().mo_normalize_using_weight_map_16s(w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), w.shape[1], w.shape[0])
    assert np.array_equal(v, g["norm16_out"])
    for i, gv in enumerate(g["gain_values"]):
        assert np.array_equal(O.gain_apply(g["gain_src"].copy(), float(gv)), g["gain_out_%d" % i]), float(gv)
    m = g["w255_src"]
    assert np.array_equal(_bits((m.astype(np.float32) * np.float32(1. / 255.)).astype(np.float32)), _bits(g["w255_out"]))
    assert (g["w255_out"][m == 255] == 1.0).all()                      # a full mask weighs exactly 1.0f


def test_oracle_distance_transform_and_nv12_are_the_references(g):
    for i in range(5):
        assert np.array_equal(_bits(O.distance_transform_l1(g["dist_mask_%d" % i])), _bits(g["dist_out_%d" % i])), i
    assert np.array_equal(O.cvt_nv12_bgr(g["nv12_src"]), g["nv12_out"])


def test_oracle_gray_is_the_references(g):
    """cvtColor(BGR2GRAY) ahead of the DP seam finder's gradients (oracle/dpseam.c: gradients(), the two lines that form `gray`): with the
    Sobel taps [-1 0 1] x [1 2 1] read off the binary, a constant-colour image pins the conversion through mo_dp_gradients' inputs --
    here the conversion itself, as the oracle writes it."""
    u = g["gray8_src"].astype(np.int64)
    assert np.array_equal(((u[..., 0] * 1868 + u[..., 1] * 9617 + u[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8), g["gray8_out"])
    F = np.float32
    f = g["gray32_src"]
    e = ((f[..., 0] * F(0.114)).astype(F) + (f[..., 1] * F(0.587)).astype(F)).astype(F)
    e = (e + (f[..., 2] * F(0.299)).astype(F)).astype(F)
    assert np.array_equal(_bits(e), _bits(g["gray32_out"]))
    # ... and the oracle's own gradients of images whose gray is linear in x: d/dx of gray = 8 * slope at every interior pixel
    ramp = np.zeros((9, 16, 3), np.uint8)
    ramp[..., 1] = np.arange(16, dtype=np.uint8)[None, :] * 8
    gx, gy = O.dp_gradients(ramp)
    gray = ((ramp[..., 1].astype(np.int64) * 9617 + (1 << 13)) >> 14).astype(np.float32)
    assert np.array_equal(gx[4, 1:-1], (gray[4, 2:] - gray[4, :-2]) * 4) and not gy[1:-1].any()


def test_oracle_float_sobel_and_gain_smoothing_are_the_references(g):
    """Sobel(CV_32F, ksize 3) through the DLL's own row / column filter objects, widths of every residue mod 8 (the SSE bodies, the
    scalar pairs, the tails): oracle/dpseam.c's position-dependent sums, confirmed by execution; and sepFilter2D([.25 .5 .25]) of the
    block-gain maps."""
    for i in range(7):
        gx, gy = O.dp_gradients(g["sobel_src_%d" % i])
        assert np.array_equal(_bits(gx), _bits(g["sobel_gx_%d" % i])) and np.array_equal(_bits(gy), _bits(g["sobel_gy_%d" % i])), i
    O.lib().mo_smooth_gain_map.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for i in range(4):
        m = g["smooth_src_%d" % i].copy()
        O.lib().mo_smooth_gain_map(m.ctypes.data_as(C.c_void_p), m.shape[1], m.shape[0])
        assert np.array_equal(_bits(m), _bits(g["smooth_out_%d" % i])), i


def test_oracle_blocks_gain_apply_is_the_references(g):
    """the reference's per-frame gain step end to end: resize (the imgproc DLL's row loop) + the exe's own compiled apply()"""
    assert np.array_equal(_bits(O.resize_linear_32f(g["bgain_map"], 136, 72)), _bits(g["bgain_full"]))
    assert np.array_equal(O.blocks_gain_apply(g["bgain_frame"].copy(), g["bgain_map"]), g["bgain_out"])


def test_oracle_linear_resize_is_the_references(g):
    """cv::resize(INTER_LINEAR): the float one of the block-gain maps, the 8-bit fixed-point one of the mask fix-up (edge rows keep both weights)"""
    for i in range(5):
        o = g["rsz32_out_%d" % i]
        assert np.array_equal(_bits(O.resize_linear_32f(g["rsz32_src_%d" % i], o.shape[1], o.shape[0])), _bits(o)), i
        o = g["rsz8_out_%d" % i]
        assert np.array_equal(O.resize_linear_8u(g["rsz8_src_%d" % i], o.shape[1], o.shape[0]), o), i


@pytest.mark.gpu
def test_engine_stages_are_the_references(g):
    """The HIP stage-level entry points on the reference's own outputs: the warp of a whole camera (maps from the projector, BORDER_REFLECT),
    the fixed-point-map remap of the lens undistort (BORDER_CONSTANT) and the scalar gain apply."""
    import stitchingvideo_amd as S
    from stitchingvideo_amd import capi
    corner, out = S.SphericalWarper(150.0).warp(g["warp_frame"], g["warp_K"], g["warp_R"], capi.INTER_LINEAR, capi.BORDER_REFLECT)
    assert tuple(corner) == tuple(g["warp_roi"][:2]) and np.array_equal(out, g["warp_out"])
    _, mask = S.SphericalWarper(150.0).warp(np.full((90, 160), 255, np.uint8), g["warp_K"], g["warp_R"], capi.INTER_NEAREST, capi.BORDER_CONSTANT)
    assert np.array_equal(mask, g["warp_mask_out"])                      # the warped all-255 mask of stitcher.cpp:278-280
    assert np.array_equal(S.remap_fixed(g["remap_src"], g["remapc_xy"], g["remapc_fxy"]), g["remapc_out"])
    for i, gv in enumerate(g["gain_values"]):
        img = g["gain_src"].copy()
        S.GainCompensator([float(gv)]).apply(0, (0, 0), img, None)
        assert np.array_equal(img, g["gain_out_%d" % i]), float(gv)
    assert np.array_equal(S.cvt_nv12_bgr(g["nv12_src"]), g["nv12_out"])
    img = g["bgain_frame"].copy()
    S.BlocksGainCompensator([g["bgain_map"]]).apply(0, (0, 0), img, None)            # the block-gain map up-sized and applied on the device
    assert np.array_equal(img, g["bgain_out"])

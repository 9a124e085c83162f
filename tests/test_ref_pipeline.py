"""The oracle against PANORAMAS AND WHOLE-FUNCTION OUTPUTS OF THE REFERENCE ITSELF (tests/golden/ref_pipeline.npz, ref_stills_full.npz:
produced by tests/make_ref_pipeline_goldens.py, which runs the reference's shipped blender, remap, resize, dilate, copyMakeBorder,
pyramids, distance transform, convertTo and the exe's gain apply in place).  CPU; bit for bit.  The engine's side of the same vectors:
tests/test_gpu_ref_pipeline.py."""
import os

import numpy as np
import pytest

import golden_util as G
import helpers as H
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
WRAP3 = dict(size=(240, 135), focal=90.0)


@pytest.fixture(scope="module")
def g():
    with np.load(os.path.join(HERE, "golden", "ref_pipeline.npz")) as z:
        return {k: z[k] for k in z.files}


def wrap3_rig(g):
    cams = [dict(K=g["wrap3/K%d" % i], R=g["wrap3/R%d" % i], size=WRAP3["size"], mask=g["wrap3/mask%d" % i], gain_map=g["wrap3/gain_map%d" % i])
            for i in range(3)]
    return dict(scale=WRAP3["focal"], warp="spherical", blend="multiband", expos="gain_blocks", cameras=cams, size=WRAP3["size"]), \
        [g["wrap3/frame%d" % i] for i in range(3)]


def test_whole_remap_is_the_references(g):
    for name, interp, border in (("linear_reflect", 1, 2), ("linear_constant", 1, 0), ("nearest_constant", 0, 0), ("linear_replicate", 1, 1)):
        assert np.array_equal(O.remap(g["remap_src"], g["remap_xmap"], g["remap_ymap"], interp, border), g["remap_out_" + name]), name


def test_whole_mask_fixup_chain_is_the_references(g):
    assert np.array_equal(O.dilate3x3(g["dilate_src"]), g["dilate_out"])
    assert np.array_equal(O.resize_linear_8u(g["dilate_src"], 517, 389), g["resize8_out"])
    assert np.array_equal(O.mask_fixup(g["fix_seam"], g["fix_mask"]), g["fix_out"])          # dilate -> resize -> & (stitcher.cpp:291-293)


def test_whole_resize_32f_is_the_references(g):
    out = O.resize_linear_32f(g["resize32_src"], 257, 189)
    assert np.array_equal(out.view(np.uint32), g["resize32_out"].view(np.uint32))


def test_whole_resize_by_factor_is_the_references(g):
    """resize(full_img, img, Size(), f, f) on CV_8UC3: what --compose_megapix does to every input (CLI:699-700)"""
    for k in range(3):
        assert np.array_equal(O.resize_linear_8u_factor(g["resizef_src"], float(g["resizef_f%d" % k])), g["resizef_out%d" % k]), k


def test_whole_copy_make_border_is_the_references(g):
    for b in (0, 1, 2, 4):
        assert np.array_equal(O.copy_make_border_16s(g["cmb_src"], 7, 22, 13, 5, b), g["cmb_out_%d" % b]), b


def test_whole_pyramids_on_long_rows_are_the_references(g):
    assert np.array_equal(O.pyr_down_16s(g["pyr_src"]), g["pyr_down_out"])
    assert np.array_equal(O.pyr_up_16s(g["pyr_up_src"]), g["pyr_up_out"])
    assert np.array_equal(O.pyr_down_32f(g["pyr32_src"]).view(np.uint32), g["pyr32_down_out"].view(np.uint32))


def test_whole_distance_transform_is_the_references(g):
    assert np.array_equal(O.distance_transform_l1(g["dist_src"]).view(np.uint32), g["dist_out"].view(np.uint32))


@pytest.mark.parametrize("name,kind,bands,wt", [("mb3_32f", O.BLEND_MULTI_BAND, 3, O.WEIGHT_32F), ("mb3_16s", O.BLEND_MULTI_BAND, 3, O.WEIGHT_16S),
                                               ("mb5_32f", O.BLEND_MULTI_BAND, 5, O.WEIGHT_32F), ("feather", O.BLEND_FEATHER, 0, O.WEIGHT_32F),
                                               ("no", O.BLEND_NO, 0, O.WEIGHT_32F)])
def test_oracle_blender_equals_the_references_blender(g, name, kind, bands, wt):
    """cv::detail::{MultiBandBlender, FeatherBlender, Blender}::prepare / feed / blend of the shipped DLL, on the same three images"""
    b = O.Blender(kind, num_bands=max(bands, 1), sharpness=0.02, weight_type=wt)
    corners = [tuple(int(v) for v in c) for c in g["blend_corners"]]
    sizes = [tuple(int(v) for v in s) for s in g["blend_sizes"]]
    b.prepare(corners, sizes)
    assert b.dst_roi()[:2] == tuple(int(v) for v in g["blend_roi"][:2])
    for i, c in enumerate(corners):
        b.feed(g["blend_img%d" % i], g["blend_mask%d" % i], c)
    if kind == O.BLEND_MULTI_BAND:
        assert b.num_bands() == int(g["blend_%s_bands" % name])
    _, dst, mask = b.blend()
    assert np.array_equal(mask, g["blend_%s_mask" % name])
    assert np.array_equal(dst, g["blend_%s_dst" % name]), H.diff_report(dst, g["blend_%s_dst" % name])
    assert np.array_equal(np.clip(dst, 0, 255).astype(np.uint8), g["blend_%s_pano" % name])         # result.convertTo(pano, CV_8U)


@pytest.mark.parametrize("name", G.CASES)
def test_committed_goldens_are_the_references_panoramas(g, name):
    """the panoramas of the older golden files (made with the oracle) ARE what the reference's compose loop returns for their inputs"""
    rig, frames, z = G.load(name)
    assert tuple(int(v) for v in z["rect"]) == tuple(int(v) for v in g[name + "/rect"])
    assert np.array_equal(z["pano_mask"], g[name + "/pano_mask"])
    assert np.array_equal(z["pano"], g[name + "/pano"])


@pytest.mark.parametrize("tag,blend,wt", [("32f", "multiband", None), ("16s", "multiband", "16s"), ("feather", "feather", None)])
def test_oracle_wraparound_rig_equals_the_reference(g, tag, blend, wt):
    rig, frames = wrap3_rig(g)
    rect, pano, mask = H.oracle_compose(rig, frames, blend=blend, num_bands=5, weight_type=wt)
    assert rect == tuple(int(v) for v in g["wrap3/rect"])
    assert rect[2] > 2 * 3.14 * WRAP3["focal"] - 4            # one camera straddles the seam: the canvas is the whole circle wide
    assert np.array_equal(mask, g["wrap3/pano_mask_" + tag])
    assert np.array_equal(pano, g["wrap3/pano_" + tag]), H.diff_report(pano, g["wrap3/pano_" + tag])


def test_oracle_full_size_stills_equal_the_reference():
    """two of the reference's bundled stills at their own 1920 x 1088: the oracle's panorama is the reference's, bit for bit"""
    rig, frames, z = G.load("ref_stills_full")
    assert frames[0].shape == (1088, 1920, 3)
    rect, pano, mask = H.oracle_compose(rig, frames, blend="multiband", num_bands=5, threads=8)
    assert rect == tuple(int(v) for v in z["rect"])
    assert np.array_equal(mask, z["pano_mask"])
    assert np.array_equal(pano, z["pano"]), H.diff_report(pano, z["pano"])


# ---- the calibration side: the reference's exposure compensators, seam finders and warpers, run whole -----------------------------------------

def _cal(g, kind):
    n = len(g["cal_sizes"])
    corners = [tuple(int(v) for v in c) for c in g["cal_corners"]]
    return corners, [g["cal_%s%d" % (kind, i)] for i in range(n)], [g["cal_mask%d" % i] for i in range(n)]


def test_oracle_gain_compensator_feed_is_the_references(g):
    """GainCompensator::feed (overlap statistics, normal equations, cv::solve LU) and BlocksGainCompensator::feed (per-block feed, gain
    maps, sepFilter2D smoothing): doubles and floats bit for bit"""
    corners, imgs, masks = _cal(g, "img8_")
    assert np.array_equal(np.asarray(O.gain_feed(corners, imgs, masks)).view(np.uint64), g["cal_gains"].view(np.uint64))
    for bw, bh in ((32, 32), (16, 24)):
        for i, m in enumerate(O.blocks_gain_feed(corners, imgs, masks, (bw, bh))):
            want = g["cal_gain_map_%dx%d_%d" % (bw, bh, i)]
            assert m.shape == want.shape and np.array_equal(m.view(np.uint32), want.view(np.uint32)), (bw, bh, i)


@pytest.mark.parametrize("kind,tag", [("voronoi", "f"), ("dp_color", "f"), ("dp_colorgrad", "f"), ("dp_color", "8"), ("dp_colorgrad", "8")])
def test_oracle_seam_finders_are_the_references(g, kind, tag):
    """VoronoiSeamFinder::find, DpSeamFinder(COLOR | COLOR_GRAD)::find on CV_32FC3 (what the sample passes) and CV_8UC3 images"""
    corners, imgs, masks = _cal(g, "imgf" if tag == "f" else "img8_")
    got = O.voronoi_seams(corners, masks) if kind == "voronoi" else O.dp_seams(corners, imgs, masks, "color" if kind == "dp_color" else "color_grad")
    changed = 0
    for i, m in enumerate(got):
        want = g["cal_seam_%s_%s_%d" % (kind, tag, i)]
        assert np.array_equal(m, want), (kind, tag, i, int((m != want).sum()))
        changed += int((want != masks[i]).sum())
    assert changed > 500                                          # the reference's finder did carve seams


MAP_WARPS = ["plane", "spherical", "cylindrical", "fisheye", "stereographic", "compressedPlaneA2B1", "compressedPlaneA1.5B1",
             "compressedPlanePortraitA2B1", "compressedPlanePortraitA1.5B1", "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1",
             "paniniPortraitA1.5B1", "mercator", "transverseMercator", "sphericalPortrait", "cylindricalPortrait", "planePortrait"]


@pytest.mark.parametrize("name", MAP_WARPS)
def test_oracle_backward_maps_are_the_references(g, name):
    """<X>Warper::buildMaps of the shipped DLL (setCameraParams, detectResultRoi, mapBackward per pixel).  plane / planePortrait: no math
    library at all -- pinned outright; the others ran with THIS host's libm answering MSVCR100's sinf / cosf / atan2f ...: formulas,
    operation order and ROI detection are the reference's, the library's function values are not pinnable."""
    from stitchingvideo_amd import _capi as capi
    roi, xm, ym = O.build_maps(capi.WARP_NAMES[name], 90.0, g["maps_K"], g["maps_R"], 96, 54)
    r = tuple(int(v) for v in g["maps_%s_roi" % name])
    assert (r[0], r[1], r[2] + 1, r[3] + 1) == tuple(roi)         # buildMaps returns Rect(tl, br); warpRoi / the maps are one larger (warpers_inl.hpp:58-74)
    assert np.array_equal(xm.view(np.uint32), g["maps_%s_x" % name].view(np.uint32))
    assert np.array_equal(ym.view(np.uint32), g["maps_%s_y" % name].view(np.uint32))


def test_oracle_lens_undistort_and_nv12_are_the_references(g):
    """initUndistortRectifyMap(K, D, Mat(), K, size, CV_16SC2) as RunTestEyeFishMap calls it (DRV:201-238), the remap on its fixed-point maps,
    and cvtColor(COLOR_YUV2BGR_NV12) -- whole functions of the shipped imgproc"""
    m1, m2 = O.init_undistort_rectify_map(g["und_K"], g["und_dist"], (320, 180))
    assert np.array_equal(m1, g["und_map1"]) and np.array_equal(m2, g["und_map2"])
    assert np.array_equal(O.remap_fixed(g["und_frame"], m1, m2), g["und_out"])
    assert np.array_equal(O.cvt_nv12_bgr(g["nv12w_src"]), g["nv12w_out"])


def test_product_host_rectify_maps_are_the_references(g):
    """the product's own host-side map builder (csrc/geometry.cpp, no GPU needed)"""
    import stitchingvideo_amd as S
    m1, m2 = S.init_undistort_rectify_map(g["und_K"], list(g["und_dist"]), (320, 180))
    assert np.array_equal(m1, g["und_map1"]) and np.array_equal(m2, g["und_map2"])

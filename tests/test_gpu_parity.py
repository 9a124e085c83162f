"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar: bit-exact -- every stage is integer/byte work, and the float parts (backward map, gain
field, weight pyramids) are evaluated in the same IEEE order on both sides.
"""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import capi, rig as rigmod

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(20260210)


def small_cam(warp, yaw=20.0, w=320, h=180, focal=210.0):
    return rigmod.intrinsics(focal, w, h), rigmod.rot_y(yaw), (w, h), focal


@pytest.mark.parametrize("warp", ["plane", "spherical", "cylindrical"])
def test_build_maps_bit_exact(warp):
    K, R, size, focal = small_cam(warp)
    warper = S.engine.WARPERS[warp](focal)
    roi, xm, ym = warper.buildMaps(size, K, R)
    oroi, ox, oy = O.build_maps(H.WARP_O[warp], focal, K, R, *size)
    assert roi == oroi
    assert np.array_equal(xm.view(np.uint32), ox.view(np.uint32))
    assert np.array_equal(ym.view(np.uint32), oy.view(np.uint32))
    _, hx, hy = warper.buildMaps(size, K, R, host=True)
    assert np.array_equal(hx.view(np.uint32), ox.view(np.uint32)) and np.array_equal(hy.view(np.uint32), oy.view(np.uint32))


@pytest.mark.parametrize("warp", ["plane", "spherical", "cylindrical"])
@pytest.mark.parametrize("yaw", [0.0, 33.0, 180.0])
def test_warp_image_and_mask_bit_exact(warp, yaw):
    if warp == "plane" and yaw == 180.0:
        pytest.skip("plane projection of a camera facing backwards is degenerate")
    K, R, (w, h), focal = small_cam(warp, yaw)
    warper = S.engine.WARPERS[warp](focal)
    img = RNG.integers(0, 256, (h, w, 3), dtype=np.uint8)
    corner, dst = warper.warp(img, K, R, capi.INTER_LINEAR, capi.BORDER_REFLECT)
    oroi, odst = O.warp(H.WARP_O[warp], focal, K, R, img, O.INTER_LINEAR, O.BORDER_REFLECT)
    assert corner == oroi[:2] and dst.shape == odst.shape
    assert np.array_equal(dst, odst), H.diff_report(dst, odst)
    mask = np.full((h, w), 255, np.uint8)
    _, mdst = warper.warp(mask, K, R, capi.INTER_NEAREST, capi.BORDER_CONSTANT)
    _, omdst = O.warp(H.WARP_O[warp], focal, K, R, mask, O.INTER_NEAREST, O.BORDER_CONSTANT)
    assert np.array_equal(mdst, omdst), H.diff_report(mdst, omdst)
    # the other two mode combinations the kernel template carries
    _, d2 = warper.warp(img, K, R, capi.INTER_LINEAR, capi.BORDER_CONSTANT)
    _, o2 = O.warp(H.WARP_O[warp], focal, K, R, img, O.INTER_LINEAR, O.BORDER_CONSTANT)
    assert np.array_equal(d2, o2), H.diff_report(d2, o2)
    _, d3 = warper.warp(img[:, :, 0].copy(), K, R, capi.INTER_NEAREST, capi.BORDER_REFLECT)
    _, o3 = O.warp(H.WARP_O[warp], focal, K, R, img[:, :, 0].copy(), O.INTER_NEAREST, O.BORDER_REFLECT)
    assert np.array_equal(d3, o3), H.diff_report(d3, o3)


def test_warp_strided_source():
    K, R, (w, h), focal = small_cam("spherical", 10.0)
    big = RNG.integers(0, 256, (h, w + 13, 3), dtype=np.uint8)
    view = big[:, 5:5 + w]                      # cv::Mat ROI: step > cols*3
    _, dst = S.SphericalWarper(focal).warp(view, K, R)
    _, odst = O.warp(O.WARP_SPHERICAL, focal, K, R, np.ascontiguousarray(view), O.INTER_LINEAR, O.BORDER_REFLECT)
    assert np.array_equal(dst, odst)


@pytest.mark.parametrize("shape", [(255, 410), (64, 64), (33, 97)])
def test_blocks_gain_apply_bit_exact(shape):
    h, w = shape
    img = RNG.integers(0, 256, (h, w, 3), dtype=np.uint8)
    gm = (1.0 + 0.3 * RNG.standard_normal(((h + 31) // 32, (w + 31) // 32))).astype(np.float32)
    got = img.copy()
    S.BlocksGainCompensator([gm]).apply(0, (0, 0), got)
    assert np.array_equal(got, O.blocks_gain_apply(img, gm)), H.diff_report(got, O.blocks_gain_apply(img, gm))
    # gain map already at image size: no resize (exposure_compensate.cpp:230-231)
    full = (1.0 + 0.2 * RNG.standard_normal((h, w))).astype(np.float32)
    got = img.copy()
    S.BlocksGainCompensator([full]).apply(0, (0, 0), got)
    assert np.array_equal(got, O.blocks_gain_apply(img, full))
    got = img.copy()
    S.GainCompensator([1.37]).apply(0, (0, 0), got)
    assert np.array_equal(got, O.gain_apply(img, 1.37))


def _rig(name, div, expos="gain_blocks"):
    return rigmod.build_rig(rigmod.scaled_config(name, div), expos=expos)


def test_multiband_intermediates_bit_exact():
    """Per-camera Gaussian pyramids, weight pyramids and dst_band_weights_ against the oracle's feed()."""
    rig = _rig("C3", 8)
    frames = rigmod.frames_for(rig, 0)
    eng = H.make_engine(rig, debug=True)
    pano, mask = eng.compose(frames)
    nb = eng.num_bands()

    def same_where_needed(got, want, i, l, what):
        # the engine only computes Gaussian pixels that can reach the panorama (work lists); the rest read 0
        need = eng.debug_read(capi.DBG_CAM_NEED_U8, i, l).astype(bool)
        assert need.any()
        bad = (got != want) & need[None]
        assert not bad.any(), "cam %d %s level %d: %d needed pixels differ" % (i, what, l, int(bad.sum()))
        return need
    # oracle: replay warp + gain + to16S, then the blender with introspection
    ob = O.Blender(O.BLEND_MULTI_BAND, 5)
    ob.prepare(rig["corners"], rig["sizes"])
    assert ob.num_bands() == nb
    canvas = ob.dst_roi()
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        _, warped = O.warp(O.WARP_SPHERICAL, rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        warped = O.blocks_gain_apply(warped, c["gain_map"])
        w16 = warped.astype(np.int16)
        g = O.multiband_feed_geometry(canvas, nb, rig["corners"][i], rig["sizes"][i])
        top, left, bottom, right = g[4:8]
        lvl = O.copy_make_border_16s(w16, top, bottom, left, right, O.BORDER_REFLECT)
        got0 = eng.debug_read(capi.DBG_CAM_WARPED_S16, i, 0)
        need0 = same_where_needed(got0, lvl.transpose(2, 0, 1), i, 0, "warped")
        wl = np.zeros(lvl.shape[:2], np.float32)
        wl[top:top + w16.shape[0], left:left + w16.shape[1]] = c["mask"].astype(np.float32) * np.float32(1.0 / 255.0)
        assert need0[wl != 0].all()              # every pixel with a level-0 weight is computed
        for l in range(1, nb + 1):
            lvl = O.pyr_down_16s(lvl)
            got = eng.debug_read(capi.DBG_CAM_GAUSS_S16, i, l)
            same_where_needed(got, lvl.transpose(2, 0, 1), i, l, "gauss")
        for l in range(0, nb + 1):
            gotw = eng.debug_read(capi.DBG_CAM_WEIGHT_F32, i, l)
            assert np.array_equal(gotw.view(np.uint32), wl.view(np.uint32)), "cam %d weight level %d" % (i, l)
            if l < nb:
                wl = O.pyr_down_32f(wl)
        ob.feed(w16, c["mask"], rig["corners"][i])
    for l in range(nb + 1):
        ws = eng.debug_read(capi.DBG_BAND_WEIGHT_F32, 0, l)
        assert np.array_equal(ws.view(np.uint32), ob.band_weight(l).view(np.uint32)), "band weight level %d" % l
    # collapsed canvas levels: normalise the oracle's accumulated bands and collapse coarse to fine
    norm = []
    for l in range(nb + 1):
        band = np.ascontiguousarray(ob.band(l))
        wt = np.ascontiguousarray(ob.band_weight(l))
        O.lib().mo_normalize_using_weight_map_32f(wt.ctypes.data, band.ctypes.data, band.shape[1], band.shape[0])
        norm.append(band)
    coll = norm[nb]
    for l in range(nb, 0, -1):
        got = eng.debug_read(capi.DBG_CANVAS_S16, 0, l)
        assert np.array_equal(got, coll.transpose(2, 0, 1)), "canvas level %d: %s" % (l, H.diff_report(got, coll.transpose(2, 0, 1)))
        up = O.pyr_up_16s(coll).astype(np.int32) + norm[l - 1].astype(np.int32)
        coll = np.clip(up, -32768, 32767).astype(np.int16)
    rect, odst, omask = ob.blend()
    assert np.array_equal(mask, omask)
    assert np.array_equal(pano, np.clip(odst, 0, 255).astype(np.uint8)), H.diff_report(pano, np.clip(odst, 0, 255))


COMPOSE_CASES = [
    ("C3", 4, "gain_blocks", 5), ("C3", 8, "gain", 5), ("C3", 8, "no", 3), ("C4", 4, "gain_blocks", 5),
    ("C5", 8, "gain_blocks", 5), ("C2", 4, "gain_blocks", 4), ("C3", 6, "gain_blocks", 7),
]


@pytest.mark.parametrize("name,div,expos,bands", COMPOSE_CASES)
def test_compose_multiband_bit_exact(name, div, expos, bands):
    rig = _rig(name, div, expos)
    frames = rigmod.frames_for(rig, 3)
    eng = H.make_engine(rig, blend="multiband", num_bands=bands)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, blend="multiband", num_bands=bands)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)


@pytest.mark.parametrize("name,div,sharp", [("C2", 4, 0.02), ("C2", 2, 0.02), ("C3", 8, 0.1)])
def test_compose_feather_bit_exact(name, div, sharp):
    rig = _rig(name, div)
    frames = rigmod.frames_for(rig, 1)
    eng = H.make_engine(rig, blend="feather", sharpness=sharp)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, blend="feather", sharpness=sharp)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)


@pytest.mark.parametrize("name,div", [("C2", 4), ("C3", 8)])
def test_compose_no_blend_bit_exact(name, div):
    rig = _rig(name, div)
    frames = rigmod.frames_for(rig, 2)
    eng = H.make_engine(rig, blend="no")
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, blend="no")
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask) and np.array_equal(pano, opano), H.diff_report(pano, opano)


def test_driver_lut_path_bit_exact():
    """The reference driver's own frame loop: remap + BlockApply + to16S + feedSizeRemap + to8U."""
    rig = _rig("C5", 8)
    frames = rigmod.frames_for(rig, 5)
    eng = H.make_engine(rig, blend="no", driver_lut=True, crop=(0.1, 0.1, 10.0, 10.0))
    pano, _ = eng.compose(frames, want_mask=False)
    imgs = []
    for c, f in zip(rig["cameras"], frames):
        _, warped = O.warp(O.WARP_CYLINDRICAL, rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        imgs.append(O.blocks_gain_apply(warped, c["gain_map"]).astype(np.int16))
    lut = O.Lut(rig["corners"], rig["sizes"], [c["mask"] for c in rig["cameras"]])
    rect, opano = lut.compose(imgs, 0.1, 0.1, 10.0, 10.0)
    assert eng.canvas() == rect
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)


def test_recalibration_swaps_state():
    """mst_update publishes a new state that the next compose uses (UpdateMat, DRV:696-722)."""
    rig_a, rig_b = _rig("C3", 8), _rig("C3", 8, "gain")
    frames = rigmod.frames_for(rig_a, 0)
    eng = S.PanoramaEngine(6, warp="spherical", blend="multiband", expos="gain_blocks")
    with pytest.raises(capi.StitchError) as ei:
        eng.compose(frames)
    assert ei.value.code == capi.ERR_NOT_CALIBRATED
    eng.update(rig_a["scale"], rig_a["cameras"])
    p1, _ = eng.compose(frames)
    cams = [dict(c) for c in rig_a["cameras"]]
    for c in cams:
        c["gain_map"] = (c["gain_map"] * np.float32(0.5)).astype(np.float32)
    eng.update(rig_a["scale"], cams)
    p2, _ = eng.compose(frames)
    rig_h = dict(rig_a, cameras=cams)
    _, o2, _ = H.oracle_compose(rig_h, frames)
    assert np.array_equal(p2, o2) and not np.array_equal(p1, p2)


def test_bad_arguments_are_errors():
    rig = _rig("C3", 8)
    eng = S.PanoramaEngine(6)
    cams = [dict(c) for c in rig["cameras"]]
    cams[2]["mask"] = cams[2]["mask"][:-1]          # not the warpRoi size -> CV_Assert-class error
    with pytest.raises(capi.StitchError) as ei:
        eng.update(rig["scale"], cams)
    assert ei.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.StitchError):
        eng.update(rig["scale"], rig["cameras"][:5])
    with pytest.raises(capi.StitchError):
        S.PanoramaEngine(0)
    with pytest.raises(capi.StitchError):
        S.PanoramaEngine(2, blend="feather", driver_lut=True)


def test_full_size_c3_against_oracle():
    """BASELINE configs[2] at full size (6x1080p, spherical, 5 bands), one frame, bit-exact."""
    rig = rigmod.build_rig(rigmod.CONFIGS["C3"])
    frames = rigmod.frames_for(rig, 0)
    eng = H.make_engine(rig)
    assert eng.canvas() == (-3925, 1453, 7852, 1020)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, threads=8)
    assert np.array_equal(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)


@pytest.mark.parametrize("name,div,bands,soft", [("C3", 8, 5, False), ("C3", 6, 3, True), ("C2", 4, 4, True), ("C4", 8, 6, False)])
def test_compose_multiband_16s_weights_bit_exact(name, div, bands, soft):
    """MultiBandBlender(weight_type = CV_16S): integer weights, `>> 8` accumulate, `(v << 8) / (w + 1)` normalise."""
    rig = _rig(name, div)
    if soft:                                # fractional weights everywhere: exercises the integer divide
        rng = np.random.default_rng(11)
        for c in rig["cameras"]:
            m = c["mask"]
            vals = rng.integers(0, 256, size=((m.shape[0] + 7) // 8, (m.shape[1] + 7) // 8)).astype(np.uint8)
            c["mask"] = np.where(m != 0, np.kron(vals, np.ones((8, 8), np.uint8))[:m.shape[0], :m.shape[1]], 0).astype(np.uint8)
    frames = rigmod.frames_for(rig, 4)
    eng = H.make_engine(rig, blend="multiband", num_bands=bands, weight_type="16s")
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, blend="multiband", num_bands=bands, weight_type="16s")
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)
    p32, _ = H.make_engine(rig, blend="multiband", num_bands=bands).compose(frames)
    assert not np.array_equal(p32, pano)        # the two weight types really are different arithmetic

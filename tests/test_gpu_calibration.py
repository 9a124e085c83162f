"""Calibration-side products on the device: the compose loop's mask fix-up
(dilate -> resize -> & warped mask: stitcher.cpp:291-294 = DRV:676-678) and FeatherBlender's createWeightMap
(distanceTransform + threshold: blenders.cpp:427-432), against the oracle, bit for bit; and the engine fed with
seam-scale masks composing the same panorama as the oracle does from the fixed-up masks."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod

pytestmark = pytest.mark.gpu


def _seam_scale_masks(rig, seam_div, rng):
    """What the seam finder hands over: one 8-bit mask per camera at seam scale (here: the full-resolution Voronoi
    mask decimated, plus a few random holes so that dilate and the bilinear resize both matter)."""
    out = []
    for c in rig["cameras"]:
        m = c["mask"]
        h, w = max(1, m.shape[0] // seam_div), max(1, m.shape[1] // seam_div)
        ys = (np.arange(h) * m.shape[0] // h)[:, None]
        xs = (np.arange(w) * m.shape[1] // w)[None, :]
        s = m[ys, xs].copy()
        s[rng.uniform(size=s.shape) < 0.03] = 0
        out.append(s)
    return out


@pytest.mark.parametrize("name,div,seam_div,blend", [("C3", 8, 4, "multiband"), ("C2", 4, 7, "feather"), ("C5", 16, 3, "no")])
def test_mask_fixup_on_device_and_compose(name, div, seam_div, blend):
    rng = np.random.default_rng(5)
    cfg = rigmod.scaled_config(name, div)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg, seams=True)
    warper = S.engine.WARPERS[rig["warp"]](rig["scale"])
    seam = _seam_scale_masks(rig, seam_div, rng)
    cams_dev, cams_ora = [], []
    for c, sm in zip(rig["cameras"], seam):
        _, full = rigmod.warped_full_mask(warper, c["size"], c["K"], c["R"])       # warp(all-255, NEAREST, CONSTANT)
        fixed = O.mask_fixup(sm, full)
        cams_ora.append(dict(c, mask=fixed))
        d = dict(c, seam_mask=sm)
        d.pop("mask")
        cams_dev.append(d)
    eng = S.PanoramaEngine(len(cams_dev), warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=4)
    eng.update(rig["scale"], cams_dev)
    for i, c in enumerate(cams_ora):
        got = eng.camera_mask(i)
        assert np.array_equal(got, c["mask"]), (i, H.diff_report(got, c["mask"]))
        assert 0 < (got != 0).mean() < 1
    frames = rigmod.frames_for(rig, 2)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(dict(rig, cameras=cams_ora), frames, blend=blend, num_bands=4)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask) and np.array_equal(pano, opano), H.diff_report(pano, opano)
    eng.close()


@pytest.mark.parametrize("sharpness", [0.02, 0.3, 1e-5])
def test_feather_weight_map_on_device(sharpness):
    """createWeightMap for hard, holey, empty and full masks (the last one has no zero pixel: the chamfer's frame of
    INT_MAX >> 2 decides, and with a tiny sharpness that value survives the threshold)."""
    rng = np.random.default_rng(9)
    cfg = rigmod.scaled_config("C2", 6)
    rig = rigmod.build_rig(cfg, seams=True)
    cams = [dict(c) for c in rig["cameras"]]
    m0 = cams[0]["mask"].copy()
    m0[rng.uniform(size=m0.shape) < 0.002] = 0
    cams[0]["mask"] = m0
    cams[1]["mask"] = np.full_like(cams[1]["mask"], 255)            # no zero pixel at all
    rig2 = dict(rig, cameras=cams)
    frames = rigmod.frames_for(rig2, 0)
    eng = H.make_engine(rig2, blend="feather", sharpness=sharpness)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig2, frames, blend="feather", sharpness=sharpness)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask) and np.array_equal(pano, opano), H.diff_report(pano, opano)
    eng.close()


def _seam_scale_inputs(name, div, rng, exposure_spread=0.25):
    """Seam-scale warped images + masks of a rig, with a different exposure per camera (what the compensators are for)."""
    rig = rigmod.build_rig(rigmod.scaled_config(name, div), seams=False)
    frames = rigmod.frames_for(rig, 1)
    images, masks, corners = [], [], []
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        g = 1.0 + exposure_spread * np.sin(1.7 * i)
        fe = np.clip(np.rint(f.astype(np.float64) * g), 0, 255).astype(np.uint8)
        _, warped = O.warp(H.WARP_O[rig["warp"]], rig["scale"], c["K"], c["R"], fe, O.INTER_LINEAR, O.BORDER_REFLECT)
        images.append(warped)
        m = c["mask"].copy()
        m[rng.uniform(size=m.shape) < 0.01] = 0                  # a few holes: the intersection test matters
        masks.append(m)
        corners.append(rig["corners"][i])
    return corners, images, masks


@pytest.mark.parametrize("name,div", [("C3", 8), ("C2", 6), ("C4", 10)])
def test_gain_compensator_feed_matches_oracle(name, div):
    """GainCompensator::feed: overlap statistics on the device (one thread per pair, the reference's summation
    order), normal equations + cv::solve on the host -- gains bit-identical to the oracle's doubles."""
    corners, images, masks = _seam_scale_inputs(name, div, np.random.default_rng(3))
    comp = S.GainCompensator()
    comp.feed(corners, images, masks)
    want = O.gain_feed(corners, images, masks)
    assert np.array_equal(np.asarray(comp.gains).view(np.uint64), np.asarray(want).view(np.uint64)), (comp.gains, want)
    assert max(comp.gains) - min(comp.gains) > 0.02               # the exposure spread was seen


@pytest.mark.parametrize("name,div,block", [("C3", 8, (32, 32)), ("C2", 6, (32, 32)), ("C5", 16, (16, 24))])
def test_blocks_gain_compensator_feed_matches_oracle(name, div, block):
    """BlocksGainCompensator::feed: every block an image (hundreds of them, thousands of overlapping pairs), LU solve,
    two smoothing passes -- float gain maps bit-identical to the oracle's; and they round-trip through apply()."""
    corners, images, masks = _seam_scale_inputs(name, div, np.random.default_rng(8))
    comp = S.BlocksGainCompensator(bl_width=block[0], bl_height=block[1])
    comp.feed(corners, images, masks)
    want = O.blocks_gain_feed(corners, images, masks, block)
    assert len(comp.gain_maps) == len(want)
    for got, w in zip(comp.gain_maps, want):
        assert got.shape == w.shape and np.array_equal(got.view(np.uint32), w.view(np.uint32))
    img = images[0].copy()
    comp.apply(0, corners[0], img)
    assert np.array_equal(img, O.blocks_gain_apply(images[0], want[0]))


@pytest.mark.parametrize("name,div", [("C3", 8), ("C2", 4), ("C4", 8), ("C5", 16)])
def test_voronoi_seam_finder_on_device(name, div):
    """VoronoiSeamFinder::find: pairs in the reference's order, masks updated in place; the device result equals the
    oracle's C restatement (which in turn equals rig.py's independent scipy one, tests/test_oracle_kat.py)."""
    rig = rigmod.build_rig(rigmod.scaled_config(name, div), seams=False)
    masks = [c["mask"].copy() for c in rig["cameras"]]
    want = O.voronoi_seams(rig["corners"], masks)
    S.VoronoiSeamFinder().find(rig["sizes"], rig["corners"], masks)
    for i, (a, b) in enumerate(zip(masks, want)):
        assert np.array_equal(a, b), (i, H.diff_report(a, b))
    cover = sum(int((m != 0).sum()) for m in masks)
    assert 0 < cover < sum(m.size for m in masks)

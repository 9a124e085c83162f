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

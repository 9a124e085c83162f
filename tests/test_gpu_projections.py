"""Every projector of detail/warpers.hpp through the engine: the fifteen --warp names of CLI:102
plus the three detail:: portrait warpers.  Stage level (buildMaps, warp image + mask) and the whole compose path
(multi-band, feather, no-blend) against the oracle, bit for bit.  The eleven non-separable ones take their backward
maps from the host (libm per pixel, like the reference); the oracle states the libm caveat."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import _capi as capi
from stitchingvideo_amd import rig as rigmod
from test_gpu_edge_cases import custom_rig, rot

pytestmark = pytest.mark.gpu

ALL = list(capi.WARP_NAMES)
POSES = [(-28.0, 3.0, -2.0), (0.0, -4.0, 1.5), (27.0, 2.0, 3.0)]


@pytest.mark.parametrize("name", ALL)
def test_compose_multiband_every_projection(name):
    rig = custom_rig(name, POSES, (208, 152), 190.0, scale=170.0)
    frames = rigmod.frames_for(rig, 3)
    eng = H.make_engine(rig, num_bands=4)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, num_bands=4)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)
    assert mask.mean() > 40          # a real panorama, not an empty canvas
    eng.close()


@pytest.mark.parametrize("name", [n for n in ALL if capi.WARP_NAMES[n] & 0xFF >= 3])
@pytest.mark.parametrize("blend", ["feather", "no"])
def test_compose_feather_and_paste_new_projections(name, blend):
    rig = custom_rig(name, POSES[:2], (160, 120), 150.0, blend=blend, expos="gain")
    frames = rigmod.frames_for(rig, 1)
    eng = H.make_engine(rig, blend=blend)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, blend=blend)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask) and np.array_equal(pano, opano), H.diff_report(pano, opano)
    eng.close()


@pytest.mark.parametrize("name", [n for n in ALL if capi.WARP_NAMES[n] & 0xFF >= 3])
def test_stage_level_maps_and_warp_new_projections(name):
    kind = capi.WARP_NAMES[name]
    w, h, focal = 176, 132, 160.0
    K = rigmod.intrinsics(focal, w, h)
    R = rot(12.0, -5.0, 4.0)
    warper = S.engine.WARPERS[name](focal)
    roi, xmap, ymap = warper.buildMaps((w, h), K, R)                 # through the device
    oroi, oxmap, oymap = O.build_maps(kind, focal, K, R, w, h)
    assert roi == oroi
    assert np.array_equal(xmap.view(np.uint32), oxmap.view(np.uint32))
    assert np.array_equal(ymap.view(np.uint32), oymap.view(np.uint32))
    img = S.synth_frame(1, 0, w, h)
    corner, warped = warper.warp(img, K, R)
    ocorner, owarped = O.warp(kind, focal, K, R, img, O.INTER_LINEAR, O.BORDER_REFLECT)
    assert corner == tuple(ocorner[:2]) and np.array_equal(warped, owarped)
    m = np.full((h, w), 255, np.uint8)
    _, wm = warper.warp(m, K, R, capi.INTER_NEAREST, capi.BORDER_CONSTANT)
    _, owm = O.warp(kind, focal, K, R, m, O.INTER_NEAREST, O.BORDER_CONSTANT)
    assert np.array_equal(wm, owm)


def test_custom_a_b_through_the_engine():
    """CompressedRectilinearWarper(scale, A, B) with an (A, B) the command line does not offer."""
    rig = custom_rig("compressedPlaneA2B1", POSES[:2], (160, 120), 150.0)
    # rebuild the rig's masks for (A, B) = (1.25, 0.8) through the oracle's custom slot
    import ctypes
    O.lib().mo_set_custom_ab(ctypes.c_float(1.25), ctypes.c_float(0.8))
    kind = (capi.WARP_COMPRESSED_PLANE | capi.WARP_AB_CUSTOM)
    w, h = rig["size"]
    cams = []
    for c in rig["cameras"]:
        roi, xm, ym = O.build_maps(kind, rig["scale"], c["K"], c["R"], w, h)
        inside = (np.rint(xm) >= 0) & (np.rint(xm) < w) & (np.rint(ym) >= 0) & (np.rint(ym) < h)
        cams.append(dict(c, roi=roi, mask=np.where(inside, 255, 0).astype(np.uint8),
                         gain_map=rigmod.block_gain_map(len(cams), roi[2], roi[3])))
    frames = rigmod.frames_for(rig, 0)
    eng = S.PanoramaEngine(len(cams), warp="compressedPlaneA2B1", blend="multiband", expos="gain_blocks", num_bands=3,
                           warp_ab=(1.25, 0.8))
    eng.update(rig["scale"], cams)
    pano, mask = eng.compose(frames)
    ocams = [dict(K=c["K"], R=c["R"], frame=f, mask=c["mask"], gain_map=c["gain_map"], gain=1.0) for c, f in zip(cams, frames)]
    rect, opano, omask = O.compose_panorama(kind, rig["scale"], O.BLEND_MULTI_BAND, O.EXPOS_GAIN_BLOCKS, ocams, num_bands=3)
    assert eng.canvas() == rect and np.array_equal(pano, opano) and np.array_equal(mask, omask)
    eng.close()

"""The eleven non-separable projectors (fisheye ... plane portrait: detail/warpers_inl.hpp:302-759) on the host:
warpRoi and buildMaps of the product's geometry module against the oracle's restatement, bit for bit -- both call
this image's libm, where the reference ran MSVC's (that caveat is the oracle's, see oracle/warpers.c)."""
import math

import numpy as np
import pytest

import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import _capi as capi
from stitchingvideo_amd import rig as rigmod

NEW = [n for n, k in capi.WARP_NAMES.items() if (k & 0xFF) >= 3]


def _pose(yaw, pitch=0.0, roll=0.0):
    y, p, r = (math.radians(v) for v in (yaw, pitch, roll))
    Ry = np.array([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
    Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]])
    return (Ry @ Rx @ Rz).astype(np.float32)


def test_every_cli_warp_name_resolves():
    """CLI:102 lists fifteen --warp values; mst_warp_kind_from_name knows them all (and only them)."""
    cli = ("plane|cylindrical|spherical|fisheye|stereographic|compressedPlaneA2B1|compressedPlaneA1.5B1|"
           "compressedPlanePortraitA2B1|compressedPlanePortraitA1.5B1|paniniA2B1|paniniA1.5B1|paniniPortraitA2B1|"
           "paniniPortraitA1.5B1|mercator|transverseMercator").split("|")
    L = capi.load()
    assert len(cli) == 15
    for name in cli:
        assert L.mst_warp_kind_from_name(name.encode()) == capi.WARP_NAMES[name], name
    assert L.mst_warp_kind_from_name(b"equirectangular") == -1


@pytest.mark.parametrize("name", NEW)
@pytest.mark.parametrize("pose", [(0.0, 0.0, 0.0), (25.0, -8.0, 5.0)])
def test_warp_roi_and_maps_match_oracle(name, pose):
    kind = capi.WARP_NAMES[name]
    w, h, focal = 96, 64, 80.0
    K = rigmod.intrinsics(focal, w, h)
    R = _pose(*pose)
    warper = S.engine.WARPERS[name](focal * 0.9)
    roi = warper.warpRoi((w, h), K, R)
    assert roi == O.warp_roi(kind, focal * 0.9, K, R, w, h)
    assert 0 < roi[2] < 4000 and 0 < roi[3] < 4000
    _, xmap, ymap = warper.buildMaps((w, h), K, R, host=True)
    oroi, oxmap, oymap = O.build_maps(kind, focal * 0.9, K, R, w, h)
    assert oroi == roi
    assert np.array_equal(xmap.view(np.uint32), oxmap.view(np.uint32))
    assert np.array_equal(ymap.view(np.uint32), oymap.view(np.uint32))
    # the map is a real one: a good share of the roi lands inside the source
    inside = (xmap >= 0) & (xmap < w) & (ymap >= 0) & (ymap < h)
    assert inside.mean() > 0.2


def test_projector_round_trip():
    """mapBackward(mapForward(p)) == p to float accuracy for every projector (a property of the maps, not of any
    implementation): evaluated through buildMaps at the forward image of a few source pixels."""
    w, h, focal = 160, 120, 140.0
    K = rigmod.intrinsics(focal, w, h)
    R = _pose(10.0, 4.0, -3.0)
    for name in capi.WARP_NAMES:
        kind = capi.WARP_NAMES[name]
        (x0, y0, rw, rh), xmap, ymap = O.build_maps(kind, focal, K, R, w, h)
        # every source pixel must be hit (within a pixel) by some destination pixel of the roi
        hit = np.zeros((h, w), bool)
        ok = (xmap > -0.5) & (xmap < w - 0.5) & (ymap > -0.5) & (ymap < h - 0.5)
        hit[np.rint(ymap[ok]).astype(int), np.rint(xmap[ok]).astype(int)] = True
        inner = hit[8:-8, 8:-8]
        assert inner.mean() > 0.5, (name, inner.mean())

"""tests/golden/round3_extras.npz (tests/golden/make_golden_r3.py): the round-3 restatements -- DP seam finder, NV12
conversion, lens-map composition -- pinned against regressions on the CPU, and reproduced by the device code."""
import os

import numpy as np
import pytest

import golden_util as G
import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S


def _z():
    return np.load(os.path.join(G.GOLDEN_DIR, "round3_extras.npz"))


def _lens_rig(z):
    n = int(z["lens_n"])
    cams, frames, maps = [], [], []
    for i in range(n):
        f = z["lens_frame%d" % i]
        cams.append(dict(K=z["lens_K%d" % i], R=z["lens_R%d" % i], size=(f.shape[1], f.shape[0]), mask=z["lens_mask%d" % i],
                         gain_map=z["lens_gain_map%d" % i], gain=1.0))
        frames.append(f)
        maps.append((z["lens_map1_%d" % i], z["lens_map2_%d" % i]))
    rig = dict(scale=float(z["lens_scale"]), warp="spherical", blend="multiband", expos="gain_blocks", cameras=cams, size=cams[0]["size"])
    return rig, frames, maps


def test_oracle_reproduces_round3_golden():
    z = _z()
    imgs = [z["dp_img%d" % i] for i in range(3)]
    masks = [np.full(im.shape[:2], 255, np.uint8) for im in imgs]
    corners = [tuple(int(v) for v in p) for p in z["dp_corners"]]
    for cost in ("color", "color_grad"):
        out = O.dp_seams(corners, imgs, masks, cost)
        for i in range(3):
            assert np.array_equal(out[i], z["dp_%s_mask%d" % (cost, i)]), (cost, i)
    assert np.array_equal(O.cvt_nv12_bgr(z["nv12"]), z["nv12_bgr"])
    rig, frames, maps = _lens_rig(z)
    cx, cy = O.compose_lens_maps(z["lens_probe_x"], z["lens_probe_y"], maps[0][0], maps[0][1])
    assert np.array_equal(cx.view(np.uint32), z["lens_probe_cx"].view(np.uint32)) and np.array_equal(cy.view(np.uint32), z["lens_probe_cy"].view(np.uint32))
    hx, hy = S.compose_lens_maps(z["lens_probe_x"], z["lens_probe_y"], maps[0][0], maps[0][1])      # the product's host code
    assert np.array_equal(hx.view(np.uint32), z["lens_probe_cx"].view(np.uint32)) and np.array_equal(hy.view(np.uint32), z["lens_probe_cy"].view(np.uint32))
    rect, pano, mask = H.oracle_compose(rig, frames, blend="multiband", num_bands=3, lens_maps=maps)
    assert rect == tuple(int(v) for v in z["lens_rect"]) and np.array_equal(pano, z["lens_pano"]) and np.array_equal(mask, z["lens_pano_mask"])


@pytest.mark.gpu
def test_device_reproduces_round3_golden():
    z = _z()
    imgs = [z["dp_img%d" % i] for i in range(3)]
    masks = [np.full(im.shape[:2], 255, np.uint8) for im in imgs]
    corners = [tuple(int(v) for v in p) for p in z["dp_corners"]]
    for cost in ("color", "color_grad"):
        out = [m.copy() for m in masks]
        S.DpSeamFinder(cost).find(imgs, corners, out)                # (masks updated in place, as DpSeamFinder::find does)
        for i in range(3):
            assert np.array_equal(out[i], z["dp_%s_mask%d" % (cost, i)]), (cost, i)
    assert np.array_equal(S.cvt_nv12_bgr(z["nv12"]), z["nv12_bgr"])
    rig, frames, maps = _lens_rig(z)
    for c, m in zip(rig["cameras"], maps):
        c["undistort"] = m
    eng = H.make_engine(rig, num_bands=3, lens_composed=True)
    pano, mask = eng.compose(frames)
    assert np.array_equal(pano, z["lens_pano"]) and np.array_equal(mask, z["lens_pano_mask"])
    eng.close()

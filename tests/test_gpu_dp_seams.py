"""DpSeamFinder (seam_finders.cpp:163-1045) on the device against the oracle's restatement: final masks bit for bit, on
the BASELINE rigs at parity-test size (C3 / 8: six spherical cameras, one wrapping the canvas; C2 / 4: two plane
cameras), for both cost functions and both image types, and on masks with holes and islands (several components per
pair: the conflict loop, single-neighbour merges and seam carving all run)."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod

pytestmark = pytest.mark.gpu


def _warped(rig, frame_index=0):
    frames = rigmod.frames_for(rig, frame_index)
    kind = H.WARP_O[rig["warp"]]
    imgs, masks = [], []
    for c, f in zip(rig["cameras"], frames):
        _, w = O.warp(kind, rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        imgs.append(w)
        masks.append(c["mask"].copy())
    return imgs, masks


@pytest.mark.parametrize("name,div", [("C3", 8), ("C2", 4)])
@pytest.mark.parametrize("cost", ["color", "color_grad"])
@pytest.mark.parametrize("as_float", [True, False])
def test_dp_seams_match_the_oracle(name, div, cost, as_float):
    rig = rigmod.build_rig(rigmod.scaled_config(name, div), seams=False)        # all-255 warped masks: the finder's input
    imgs, masks = _warped(rig)
    if as_float:
        imgs = [im.astype(np.float32) for im in imgs]                          # images_warped_f, CLI:488-489
    want = O.dp_seams(rig["corners"], imgs, masks, cost)
    got = [m.copy() for m in masks]
    S.DpSeamFinder(cost).find(imgs, rig["corners"], got)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (i, int((g != w).sum()))
    # the finder did something and left no pixel to two cameras
    assert any((g != m).any() for g, m in zip(got, masks))


def test_dp_seams_with_holes_and_islands():
    rng = np.random.default_rng(11)
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 6), seams=False)
    imgs, masks = _warped(rig, 3)
    for m in masks:                                   # punch holes and cut islands: many components per pair
        h, w = m.shape
        for _ in range(6):
            x, y, r = int(rng.integers(0, w)), int(rng.integers(0, h)), int(rng.integers(3, 14))
            m[max(0, y - r):y + r, max(0, x - r):x + r] = 0
        x = int(rng.integers(w // 4, 3 * w // 4))
        m[:, x:x + 2] = 0
    imgs = [im.astype(np.float32) for im in imgs]
    for cost in ("color", "color_grad"):
        want = O.dp_seams(rig["corners"], imgs, masks, cost)
        got = [m.copy() for m in masks]
        S.DpSeamFinder(cost).find(imgs, rig["corners"], got)
        for i, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), (cost, i, int((g != w).sum()))


def test_dp_seams_feed_the_engine():
    """The finder's masks as the engine's calibration input: the panorama equals the oracle's for the same masks."""
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8), seams=False)
    imgs, masks = _warped(rig)
    got = [m.copy() for m in masks]
    S.DpSeamFinder("color").find([im.astype(np.float32) for im in imgs], rig["corners"], got)
    cams = [dict(c, mask=m) for c, m in zip(rig["cameras"], got)]
    rig2 = dict(rig, cameras=cams)
    frames = rigmod.frames_for(rig2, 1)
    eng = H.make_engine(rig2, num_bands=4)
    pano, mask = eng.compose(frames)
    _, opano, omask = H.oracle_compose(rig2, frames, num_bands=4)
    assert np.array_equal(pano, opano) and np.array_equal(mask, omask)
    eng.close()

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


def _float_case(seed):
    """2 - 4 CV_32F images with NON-integer values (the association of Sobel's three-term sums matters) and coarse content (costs
    tie often: the DP's tie order decides), widths over every residue mod 8, ragged masks."""
    rng = np.random.default_rng(9100 + seed)
    n = 2 + seed % 3
    imgs, masks, corners = [], [], []
    x = 0
    for i in range(n):
        w, h = int(rng.integers(24, 70)), int(rng.integers(20, 48))
        corners.append((x + int(rng.integers(-3, 4)), int(rng.integers(-6, 7))))
        x += int(w * rng.uniform(0.45, 0.8))
        levels = 3 if seed % 2 == 0 else 17
        base = rng.integers(0, levels, size=(h, w, 3)).astype(np.float32) * np.float32(255.0 / (levels - 1) + 0.3)
        if seed % 3 == 0:
            base += rng.uniform(-0.4, 0.4, size=base.shape).astype(np.float32)
        m = np.full((h, w), 255, np.uint8)
        for yy in range(h):
            m[yy, :max(int(2 + 2 * np.sin(yy / 4.0 + i)), 0)] = 0
            m[yy, w - max(int(2 + 2 * np.cos(yy / 5.0)), 0):] = 0
        imgs.append(base.astype(np.float32)); masks.append(m)
    return corners, imgs, masks


@pytest.mark.parametrize("block", range(10))
def test_dp_colorgrad_float_images_tie_heavy(block):
    """100 random CV_32F COLOR_GRAD cases (ten per test): non-integer gray levels, so Sobel's [1 2 1] sums take the association of
    2.4.11's SSE body / scalar pairs / unrolled fours / tail by column position in k_dp_sobel and in the oracle alike; coarse
    content, so equal costs are common and the recurrence's tie order decides the seam."""
    carved = 0
    for seed in range(10 * block, 10 * block + 10):
        corners, imgs, masks = _float_case(seed)
        want = O.dp_seams(corners, imgs, masks, "color_grad")
        got = [m.copy() for m in masks]
        S.DpSeamFinder("color_grad").find(imgs, corners, got)
        for i, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), (seed, i, int((g != w).sum()))
        carved += int(any((g != m).any() for g, m in zip(got, masks)))
    assert carved >= 5

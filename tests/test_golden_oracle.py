"""The oracle against the committed golden fixtures (CPU): pins the restatement against regressions."""
import numpy as np
import pytest

import golden_util as G
import helpers as H
import oracle_lib as O


@pytest.mark.parametrize("name", G.CASES)
def test_oracle_reproduces_golden(name):
    rig, frames, z = G.load(name)
    rect, pano, mask = H.oracle_compose(rig, frames, blend=rig["blend"], num_bands=int(z["num_bands"]),
                                        sharpness=float(z["sharpness"]))
    assert rect == tuple(int(v) for v in z["rect"])
    assert np.array_equal(mask, z["pano_mask"])
    assert np.array_equal(pano, z["pano"])


def test_oracle_thread_count_does_not_change_results():
    rig, frames, z = G.load("c3_div8_multiband")
    _, pano, mask = H.oracle_compose(rig, frames, threads=4)
    assert np.array_equal(pano, z["pano"]) and np.array_equal(mask, z["pano_mask"])


def test_oracle_lut_golden():
    rig, frames, z = G.load("c5_div16_paste_lut")
    imgs, corners, sizes = [], [], []
    for c, f in zip(rig["cameras"], frames):
        roi, w = O.warp(O.WARP_CYLINDRICAL, rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        imgs.append(O.blocks_gain_apply(w, c["gain_map"]).astype(np.int16))
        corners.append(roi[:2]); sizes.append(roi[2:])
    lut = O.Lut(corners, sizes, [c["mask"] for c in rig["cameras"]])
    rect, pano = lut.compose(imgs, *[float(v) for v in z["crop"]])
    assert rect == tuple(int(v) for v in z["lut_rect"]) and np.array_equal(pano, z["lut_pano"])

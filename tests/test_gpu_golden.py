"""The HIP path against the committed golden fixtures (inputs + expected panorama), through the C ABI."""
import numpy as np
import pytest

import golden_util as G
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", G.CASES)
def test_engine_reproduces_golden(name):
    rig, frames, z = G.load(name)
    eng = H.make_engine(rig, blend=rig["blend"], num_bands=int(z["num_bands"]), sharpness=float(z["sharpness"]))
    pano, mask = eng.compose(frames)
    assert eng.canvas() == tuple(int(v) for v in z["rect"])
    assert np.array_equal(mask, z["pano_mask"]), H.diff_report(mask, z["pano_mask"])
    assert np.array_equal(pano, z["pano"]), H.diff_report(pano, z["pano"])
    eng.close()


def test_engine_driver_lut_golden():
    rig, frames, z = G.load("c5_div16_paste_lut")
    eng = H.make_engine(rig, blend="no", driver_lut=True, crop=tuple(float(v) for v in z["crop"]))
    pano, _ = eng.compose(frames, want_mask=False)
    assert eng.canvas() == tuple(int(v) for v in z["lut_rect"])
    assert np.array_equal(pano, z["lut_pano"]), H.diff_report(pano, z["lut_pano"])
    eng.close()

"""The HIP engine, through the C ABI, against PANORAMAS THE REFERENCE'S OWN CODE PRODUCED (tests/golden/ref_pipeline.npz,
ref_stills_full.npz; see tests/make_ref_pipeline_goldens.py): bit for bit."""
import os

import numpy as np
import pytest

import golden_util as G
import helpers as H
from test_ref_pipeline import wrap3_rig

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    with np.load(os.path.join(HERE, "golden", "ref_pipeline.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", G.CASES)
def test_engine_equals_the_references_panorama(g, name):
    rig, frames, z = G.load(name)
    eng = H.make_engine(rig, blend=rig["blend"], num_bands=int(z["num_bands"]), sharpness=float(z["sharpness"]))
    pano, mask = eng.compose(frames)
    assert eng.canvas() == tuple(int(v) for v in g[name + "/rect"])
    assert np.array_equal(mask, g[name + "/pano_mask"]), H.diff_report(mask, g[name + "/pano_mask"])
    assert np.array_equal(pano, g[name + "/pano"]), H.diff_report(pano, g[name + "/pano"])
    eng.close()


@pytest.mark.parametrize("tag,blend,wt", [("32f", "multiband", "32f"), ("16s", "multiband", "16s"), ("feather", "feather", "32f")])
def test_engine_wraparound_rig_equals_the_reference(g, tag, blend, wt):
    rig, frames = wrap3_rig(g)
    eng = H.make_engine(rig, blend=blend, num_bands=5, weight_type=wt)
    pano, mask = eng.compose(frames)
    assert eng.canvas() == tuple(int(v) for v in g["wrap3/rect"])
    assert np.array_equal(mask, g["wrap3/pano_mask_" + tag]), H.diff_report(mask, g["wrap3/pano_mask_" + tag])
    assert np.array_equal(pano, g["wrap3/pano_" + tag]), H.diff_report(pano, g["wrap3/pano_" + tag])
    eng.close()


def test_engine_full_size_stills_equal_the_reference():
    """two of the reference's bundled stills at their own 1920 x 1088, spherical, 5-band multi-band"""
    rig, frames, z = G.load("ref_stills_full")
    eng = H.make_engine(rig, blend="multiband", num_bands=5)
    pano, mask = eng.compose(frames)
    assert eng.canvas() == tuple(int(v) for v in z["rect"])
    assert np.array_equal(mask, z["pano_mask"]), H.diff_report(mask, z["pano_mask"])
    assert np.array_equal(pano, z["pano"]), H.diff_report(pano, z["pano"])
    eng.close()

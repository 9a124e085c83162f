"""The HIP engine, through the C ABI, against PANORAMAS THE REFERENCE'S OWN CODE PRODUCED (tests/golden/ref_pipeline.npz,
ref_stills_full.npz; see tests/make_ref_pipeline_goldens.py): bit for bit."""
import os

import numpy as np
import pytest

import golden_util as G
import helpers as H
from test_ref_pipeline import MAP_WARPS, _cal, wrap3_rig

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


# ---- the calibration-side entry points of the product against the reference's own compensators, seam finders and warpers ---------------

def test_device_gain_feeds_equal_the_reference(g):
    import stitchingvideo_amd as S
    corners, imgs, masks = _cal(g, "img8_")
    comp = S.GainCompensator()
    comp.feed(corners, imgs, masks)
    assert np.array_equal(np.asarray(comp.gains).view(np.uint64), g["cal_gains"].view(np.uint64))
    for bw, bh in ((32, 32), (16, 24)):
        comp = S.BlocksGainCompensator(bl_width=bw, bl_height=bh)
        comp.feed(corners, imgs, masks)
        for i, m in enumerate(comp.gain_maps):
            want = g["cal_gain_map_%dx%d_%d" % (bw, bh, i)]
            assert m.shape == want.shape and np.array_equal(m.view(np.uint32), want.view(np.uint32)), (bw, bh, i)


@pytest.mark.parametrize("kind,tag", [("voronoi", "f"), ("dp_color", "f"), ("dp_colorgrad", "f"), ("dp_color", "8"), ("dp_colorgrad", "8")])
def test_device_seam_finders_equal_the_reference(g, kind, tag):
    import stitchingvideo_amd as S
    corners, imgs, masks = _cal(g, "imgf" if tag == "f" else "img8_")
    got = [m.copy() for m in masks]
    if kind == "voronoi":
        S.VoronoiSeamFinder().find([(m.shape[1], m.shape[0]) for m in masks], corners, got)
    else:
        S.DpSeamFinder("color" if kind == "dp_color" else "color_grad").find(imgs, corners, got)
    for i, m in enumerate(got):
        want = g["cal_seam_%s_%s_%d" % (kind, tag, i)]
        assert np.array_equal(m, want), (kind, tag, i, int((m != want).sum()))


@pytest.mark.parametrize("name", MAP_WARPS)
def test_engine_backward_maps_equal_the_reference(g, name):
    from stitchingvideo_amd.engine import WARPERS
    roi, xm, ym = WARPERS[name](90.0).buildMaps((96, 54), g["maps_K"], g["maps_R"])
    r = tuple(int(v) for v in g["maps_%s_roi" % name])
    assert (r[0], r[1], r[2] + 1, r[3] + 1) == tuple(roi)
    assert np.array_equal(xm.view(np.uint32), g["maps_%s_x" % name].view(np.uint32))
    assert np.array_equal(ym.view(np.uint32), g["maps_%s_y" % name].view(np.uint32))


def test_device_lens_undistort_and_nv12_equal_the_reference(g):
    import stitchingvideo_amd as S
    assert np.array_equal(S.remap_fixed(g["und_frame"], g["und_map1"], g["und_map2"]), g["und_out"])
    assert np.array_equal(S.cvt_nv12_bgr(g["nv12w_src"]), g["nv12w_out"])

"""CPU-side tests of the product library: it loads, exports every symbol the header declares, its
host-only entry points (geometry, host map builder, synthetic frames) agree with the oracle, and
compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import capi, rig as rigmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    declared = set()
    for h in ("mi355_stitch.h", "mi355_ingest.h"):            # every header under include/
        header = open(os.path.join(ROOT, "include", h)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)    # (prose in comments mentions calls too)
        declared |= set(re.findall(r"\b(mst_[a-z_0-9]+)\s*\(", header))
    declared -= {"mst_config", "mst_camera", "mst_calibration", "mst_rect", "mst_engine", "mst_shm", "mst_frame_packet", "mst_build"}
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_shipped_library_is_a_product_build():
    """No -DMST_ABLATE_* (timing experiments: wrong pixels) and no -DMST_STAMPS in the library the tests and the bench load."""
    info = capi.build_info()
    assert info["timing_build"] == 0, info
    assert info["level0_packed"] in (0, 1)
    assert "TIMING" not in info["description"] and "STAMPS" not in info["description"], info
    bad = capi.Build()
    bad.struct_size = 8
    assert capi.load().mst_build_info(C.byref(bad)) == capi.ERR_INVALID_ARG


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(capi.Build) == 16 + 160
    # the C side checks struct_size; these are the sizes an LP64 compiler produces for the header
    assert C.sizeof(capi.Config) == 80          # 15 x 4-byte fields + warp_a, warp_b + strip_index, strip_count + overlap_depth
    assert C.sizeof(capi.Camera) == 8 + 36 + 36 + 12 + 4 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 16 + 8 + 8 + 8
    assert C.sizeof(capi.Calibration) == 24


@pytest.mark.parametrize("warp,okind", [("plane", O.WARP_PLANE), ("spherical", O.WARP_SPHERICAL),
                                        ("cylindrical", O.WARP_CYLINDRICAL)])
@pytest.mark.parametrize("yaw", [-77.0, 0.0, 15.0, 123.0, 180.0])
def test_warp_roi_and_host_maps_match_oracle(warp, okind, yaw):
    if warp == "plane" and abs(yaw) > 60:
        pytest.skip("plane warp of a camera looking away from the plane")
    w, h, f = 640, 360, 417.0
    K, R = rigmod.intrinsics(f, w, h), rigmod.rot_y(yaw)
    warper = S.engine.WARPERS[warp](f)
    roi = warper.warpRoi((w, h), K, R)
    assert roi == O.warp_roi(okind, f, K, R, w, h)
    _, xm, ym = warper.buildMaps((w, h), K, R, host=True)
    _, ox, oy = O.build_maps(okind, f, K, R, w, h)
    assert np.array_equal(xm.view(np.uint32), ox.view(np.uint32))
    assert np.array_equal(ym.view(np.uint32), oy.view(np.uint32))


def test_warp_roi_tilted_camera_and_plane_translation():
    # pitch + roll exercise the full 3x3 products and the spherical pole test
    a, b = np.deg2rad(80.0), np.deg2rad(7.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    R = (Rx @ Rz).astype(np.float32)
    K = rigmod.intrinsics(300.0, 400, 300, aspect=1.02)
    assert S.SphericalWarper(311.0).warpRoi((400, 300), K, R) == O.warp_roi(O.WARP_SPHERICAL, 311.0, K, R, 400, 300)
    assert S.CylindricalWarper(311.0).warpRoi((400, 300), K, rigmod.rot_y(5)) == \
        O.warp_roi(O.WARP_CYLINDRICAL, 311.0, K, rigmod.rot_y(5), 400, 300)
    T = (0.1, -0.05, 0.2)
    assert S.PlaneWarper(300.0).warpRoi((400, 300), K, rigmod.rot_y(10), T) == \
        O.warp_roi(O.WARP_PLANE, 300.0, K, rigmod.rot_y(10), 400, 300, T)
    _, xm, ym = S.PlaneWarper(300.0).buildMaps((400, 300), K, rigmod.rot_y(10), T, host=True)
    _, ox, oy = O.build_maps(O.WARP_PLANE, 300.0, K, rigmod.rot_y(10), 400, 300, T)
    assert np.array_equal(xm.view(np.uint32), ox.view(np.uint32)) and np.array_equal(ym.view(np.uint32), oy.view(np.uint32))


def test_multiband_geometry_matches_oracle_and_appendix_b():
    canvas = (-3925, 1453, 7852, 1020)
    pad, nb, out = S.multiband_geometry(canvas, 5, (-818, 1453, 1637, 1020))
    assert pad == (-3925, 1453, 7872, 1024) and nb == 5
    assert out == [-917, 1453, 939, 2477, 0, 99, 4, 120]
    rng = np.random.default_rng(3)
    for _ in range(200):
        cw, ch = int(rng.integers(40, 3000)), int(rng.integers(40, 1500))
        cx, cy = int(rng.integers(-2000, 2000)), int(rng.integers(-2000, 2000))
        bands = int(rng.integers(1, 8))
        iw, ih = int(rng.integers(1, cw + 1)), int(rng.integers(1, ch + 1))
        ix, iy = cx + int(rng.integers(0, cw - iw + 1)), cy + int(rng.integers(0, ch - ih + 1))
        pad, nb, out = S.multiband_geometry((cx, cy, cw, ch), bands, (ix, iy, iw, ih))
        ob = O.Blender(O.BLEND_MULTI_BAND, bands)
        ob.prepare([(cx, cy)], [(cw, ch)])
        assert ob.num_bands() == nb and ob.dst_roi() == pad
        assert out == O.multiband_feed_geometry(pad, nb, (ix, iy), (iw, ih))
        assert out[4] >= 0 and out[5] >= 0 and out[6] >= 0 and out[7] >= 0
        assert (out[2] - out[0]) % (1 << nb) == 0 and (out[3] - out[1]) % (1 << nb) == 0


def test_synth_frame_is_deterministic_and_in_spec():
    a = S.synth_frame(2, 7, 96, 40)
    b = S.synth_frame(2, 7, 96, 40)
    c = S.synth_frame(2, 8, 96, 40)
    d = S.synth_frame(3, 7, 96, 40)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)
    x, y = np.meshgrid(np.arange(96), np.arange(40))
    base = 128 + 80 * np.sin(2 * np.pi * (x / 97 + 2 / 7)) * np.cos(2 * np.pi * y / 61)
    for ch in range(3):
        ideal = np.clip(base + 20 * ch, 0, 255)
        err = a[:, :, ch].astype(float) - ideal
        unclipped = (ideal > 17) & (ideal < 238)
        assert np.abs(err[unclipped]).max() <= 16.5 and abs(err[unclipped].mean()) < 1.0


def test_rig_masks_partition_the_overlaps():
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    cx, cy, cw, ch = O.lib().mo_result_roi((C.c_int * 12)(*[v for p in rig["corners"] for v in p]),
                                           (C.c_int * 12)(*[v for p in rig["sizes"] for v in p]), 6).tuple()
    cover = np.zeros((ch, cw), np.int32)
    for (x, y), c in zip(rig["corners"], rig["cameras"]):
        m = c["mask"]
        cover[y - cy:y - cy + m.shape[0], x - cx:x - cx + m.shape[1]] += (m != 0)
    assert cover.max() == 1                      # Voronoi seams leave no doubly-owned pixel
    assert (cover[10:-10] == 1).mean() > 0.99    # and the ring is covered


def test_compute_entry_points_fail_loudly_without_a_gpu():
    if _has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(capi.StitchError) as ei:
        S.PanoramaEngine(2)
    assert ei.value.code == capi.ERR_NO_DEVICE
    K, R = rigmod.intrinsics(200.0, 320, 180), rigmod.rot_y(0)
    with pytest.raises(capi.StitchError) as ei:
        S.SphericalWarper(200.0).warp(np.zeros((180, 320, 3), np.uint8), K, R)
    assert ei.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.StitchError):
        S.SphericalWarper(200.0).buildMaps((320, 180), K, R)            # device variant


def test_argument_validation_without_a_gpu():
    L = capi.load()
    assert L.mst_create(None, None) == capi.ERR_INVALID_ARG
    cfg = capi.Config()
    h = C.c_void_p()
    assert L.mst_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG        # struct_size == 0
    assert b"struct_size" in L.mst_last_error(None)
    cfg.struct_size = C.sizeof(capi.Config)
    cfg.num_cameras = 99
    assert L.mst_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG
    cfg.num_cameras, cfg.warp_kind = 2, 14                 # kinds 0..13 exist (all of detail/warpers.hpp)
    assert L.mst_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG
    cfg.warp_kind, cfg.blend_kind = 1, 5
    assert L.mst_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG
    assert b"unsupported blending method" in L.mst_last_error(None)              # CV_Error text, blenders.cpp:60
    cfg.blend_kind, cfg.overlap_depth = 2, 9                                     # calls in flight: 0 .. 8
    assert L.mst_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG
    assert b"overlap_depth" in L.mst_last_error(None)
    cfg.overlap_depth, cfg.flags = 0, 64             # (round 4's MST_FLAG_STRIP_HALO: removed in round 5, refused loudly)
    assert L.mst_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG
    assert b"strip halo exchange" in L.mst_last_error(None)
    cfg.flags = 0
    r = capi.Rect()
    assert L.mst_canvas(None, C.byref(r)) == capi.ERR_INVALID_ARG
    assert L.mst_update(None, None) == capi.ERR_INVALID_ARG
    assert L.mst_compose(None, None, None, None, 0, None, 0) == capi.ERR_INVALID_ARG
    L.mst_destroy(None)

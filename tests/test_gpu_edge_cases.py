"""GPU parity on the inputs that leave the common path: tilted cameras (BORDER_REFLECT folding on every
side, spherical poles), stretched warps (map-table fallback), soft / overlapping / empty masks
(weights that do not sum to one), extreme and non-finite gains, fine gain maps, one camera, tiny sources.
Every case: engine panorama + mask == oracle panorama + mask, bit for bit."""
import math

import numpy as np
import pytest

import helpers as H
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod

pytestmark = pytest.mark.gpu


def rot(yaw, pitch=0.0, roll=0.0):
    y, p, r = (math.radians(v) for v in (yaw, pitch, roll))
    Ry = np.array([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
    Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]])
    return (Ry @ Rx @ Rz).astype(np.float32)


def custom_rig(warp, poses, size, focal, scale=None, expos="gain_blocks", seams=True, blend="multiband", block=32):
    w, h = size
    scale = float(focal if scale is None else scale)
    warper = S.engine.WARPERS[warp](scale)
    K = rigmod.intrinsics(focal, w, h)
    cams, corners, masks = [], [], []
    for pose in poses:
        R = rot(*pose)
        roi, mask = rigmod.warped_full_mask(warper, (w, h), K, R)
        cams.append(dict(K=K, R=R, size=(w, h), roi=roi))
        corners.append((roi[0], roi[1]))
        masks.append(mask)
    if seams:
        rigmod.voronoi_seams(corners, masks)
    for i, c in enumerate(cams):
        c["mask"] = masks[i]
        if expos == "gain_blocks":
            c["gain_map"] = rigmod.block_gain_map(i, c["roi"][2], c["roi"][3], block=block)
        elif expos == "gain":
            c["gain"] = 1.0 + 0.05 * math.sin(1.0 + i)
    return dict(scale=scale, warp=warp, blend=blend, expos=expos, cameras=cams, corners=corners,
                sizes=[(c["roi"][2], c["roi"][3]) for c in cams], size=(w, h))


def check(rig, blend=None, bands=4, frame=2, sharpness=0.02):
    frames = rigmod.frames_for(rig, frame)
    eng = H.make_engine(rig, blend=blend, num_bands=bands, sharpness=sharpness)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, blend=blend, num_bands=bands, sharpness=sharpness)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)
    eng.close()


@pytest.mark.parametrize("warp", ["spherical", "cylindrical", "plane"])
def test_tilted_cameras_fold_every_border(warp):
    poses = [(-22.0, 18.0, 9.0), (3.0, -20.0, -12.0), (25.0, 7.0, 30.0)]
    if warp == "spherical":
        poses.append((60.0, 70.0, 5.0))                 # looks past the pole: rays with z <= 0
    check(custom_rig(warp, poses, (200, 150), 170.0))


def test_stretched_warp_exceeds_the_table_window():
    # scale << focal: one destination pixel spans ~10 source pixels, so a 128-pixel tile row covers more
    # source columns than a table entry can address relative to its tile origin -> per-tap fallback
    check(custom_rig("plane", [(-4.0, 0.0, 0.0), (5.0, 1.0, 0.0)], (1500, 90), 700.0, scale=70.0), bands=3)
    check(custom_rig("cylindrical", [(0.0, 0.0, 2.0), (12.0, 0.0, 0.0)], (1400, 120), 900.0, scale=80.0), bands=2)


def test_soft_overlapping_and_empty_masks():
    rig = custom_rig("spherical", [(-20.0, 0, 0), (0.0, 3.0, 0), (20.0, 0, 2.0), (40.0, 0, 0)], (220, 160), 190.0, seams=False)
    rng = np.random.default_rng(7)
    levels = np.array([0, 1, 64, 127, 128, 254, 255], np.uint8)
    for i, c in enumerate(rig["cameras"]):
        m = c["mask"]
        if i == 3:
            m[:] = 0                                        # a camera that contributes nothing
            continue
        blocks = levels[rng.integers(0, len(levels), size=((m.shape[0] + 15) // 16, (m.shape[1] + 15) // 16))]
        soft = np.kron(blocks, np.ones((16, 16), np.uint8))[:m.shape[0], :m.shape[1]]
        c["mask"] = np.where(m != 0, soft, 0).astype(np.uint8)
    # complementary 127/128 pair: weights sum to exactly 1.0f from two cameras
    a, b = rig["cameras"][0]["mask"], rig["cameras"][1]["mask"]
    a[:40, :] = np.where(a[:40, :] != 0, 127, 0)
    b[:40, :] = np.where(b[:40, :] != 0, 128, 0)
    check(rig, bands=5)
    check(rig, blend="feather")
    check(rig, blend="no")


def test_extreme_and_non_finite_gains():
    rig = custom_rig("spherical", [(-15.0, 0, 0), (15.0, 0, 0)], (200, 140), 180.0)
    g = rig["cameras"][0]["gain_map"]
    g[0, 0] = 1.0e7; g[1, 2] = 3.0e38; g[2, 1] = -4.0; g[3, 3] = 0.0; g[-1, -1] = np.inf; g[2, 4] = np.nan
    g1 = rig["cameras"][1]["gain_map"]
    g1[:] = 2.5                                             # saturates most pixels
    check(rig, bands=3)
    rig2 = custom_rig("cylindrical", [(-10.0, 0, 0), (10.0, 0, 0)], (160, 120), 150.0, expos="gain")
    rig2["cameras"][0]["gain"] = 9.0e6                      # scalar gain past the bounded-rounding limit
    rig2["cameras"][1]["gain"] = 0.37
    check(rig2, bands=3)


def test_fine_gain_map_spans_many_rows_per_tile():
    check(custom_rig("spherical", [(-12.0, 4.0, 0), (14.0, -3.0, 0)], (240, 180), 200.0, block=4), bands=4)


@pytest.mark.parametrize("bands", [1, 2, 6])
def test_single_camera_and_band_counts(bands):
    check(custom_rig("spherical", [(0.0, 5.0, -3.0)], (260, 200), 210.0), bands=bands)


def test_tiny_sources_have_no_window_path():
    # 3 x 2 and 1 x 1 pixel sources: no 2x2 window can be read with the 12-byte loads, every sample is per-tap
    check(custom_rig("plane", [(0.0, 0, 0), (8.0, 0, 0)], (3, 2), 40.0, scale=400.0, expos="no"), bands=2)
    check(custom_rig("plane", [(0.0, 0, 0)], (1, 1), 30.0, scale=300.0, expos="no"), bands=1)


@pytest.mark.parametrize("blend,lut", [("feather", False), ("no", False), ("no", True), ("multiband", False)])
@pytest.mark.parametrize("pad,shift", [(1, 1), (2, 3), (0, 2)])
def test_device_frames_off_the_dword_grid(blend, lut, pad, shift):
    """Frames whose base address or row stride is not a multiple of four (a tight cv::Mat of an odd width, a view
    into a larger buffer): the staged form of the single-pass composites reads whole dwords and must hand such calls
    to the gathering / general kernels; the multi-band warp kernel shifts its funnel.  Batched device call, 2 sets."""
    import torch
    cfg = rigmod.scaled_config("C3", 8)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    n = len(rig["cameras"])
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=4, driver_lut=lut, max_batch=2)
    eng.update(rig["scale"], rig["cameras"])
    (_, _, cw, ch), _ = eng.acquire()
    dev = torch.device("cuda", 0)
    sets, ptrs, strides, keep = [], [], [], []
    for f in range(2):
        frames = rigmod.frames_for(rig, f)
        sets.append(frames)
        p, s = [], []
        for fr in frames:
            h, w, _ = fr.shape
            stride = w * 3 + pad
            buf = torch.zeros(h * stride + shift + 16, dtype=torch.uint8, device=dev)
            view = buf[shift:shift + h * stride].view(h, stride)
            view[:, :w * 3] = torch.from_numpy(fr.reshape(h, w * 3)).to(dev)
            keep.append(buf)
            p.append(buf.data_ptr() + shift)
            s.append(stride)
        ptrs.append(p)
        strides.append(s)
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()          # the fills run on torch's stream, the engine's stream does not wait for it
    eng.compose_device_batch(ptrs, strides, [p.data_ptr() for p in panos], cw * 3)
    torch.cuda.synchronize()
    for f in range(2):
        if lut:
            ref, _ = _lut_reference(rig, sets[f])
        else:
            _, ref, _ = H.oracle_compose(rig, sets[f], blend=blend, num_bands=4)
        got = panos[f].cpu().numpy()
        assert np.array_equal(got, ref), (f, H.diff_report(got, ref))
    eng.close()


def _lut_reference(rig, frames):
    """The aligned engine itself on host buffers is the reference for the LUT crop (its parity with the oracle is
    covered by test_driver_lut_path_bit_exact)."""
    n = len(rig["cameras"])
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend="no", expos=rig["expos"], driver_lut=True)
    eng.update(rig["scale"], rig["cameras"])
    out = eng.compose(frames)
    eng.close()
    return out

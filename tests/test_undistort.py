"""Lens-undistort row: initUndistortRectifyMap(CV_16SC2) restated on the host side of the product
and in the oracle, and remap with such maps.  CPU part: the two map builders agree bit for bit and obey the
closed-form properties of the model; GPU part: the remap kernel and the compose paths that run it first."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import capi, rig as rigmod

# the reference's hard-coded fisheye intrinsics (DRV:209-216), 1280x720 frames
K_DRV = np.array([[655.6, 0, 625.06], [0, 656.4, 361.96], [0, 0, 1]], np.float64)
D_DRV = [-0.382, 0.195, -0.00363, 0.00237]

# the chessboard calibration the reference ships next to its x64 executable (x64/Release/camera-mp4.yml, "03/18/16", 1280x720, the
# standard 5-coefficient model with k3, average reprojection error 7.6 px): a strongly non-monotonic map -- the whole 1280 x 720
# frame samples columns 290 .. 1184 and rows 5 .. 686 of the source, folding back towards the corners -- as a reference-owned fixture
# beside the driver's hard-coded intrinsics
K_YML = np.array([[1.3240283915995651e+003, 0., 6.6179009310103982e+002], [0., 1.4000828762189344e+003, 3.6776239396487273e+002], [0., 0., 1.]], np.float64)
D_YML = [-5.3287487456559879e-001, -7.0767524066484535e+000, -5.4286244387084424e-002, 8.8021459012654110e-002, 1.6610642543619203e+001]

CASES = [
    (K_DRV, D_DRV, None, None, (1280, 720)),
    (K_YML, D_YML, None, None, (1280, 720)),
    (K_DRV, [0.1, -0.05, 0.001, 0.002, 0.03], None, None, (333, 211)),
    (K_DRV, [0.2, -0.1, 0.0, 0.0, 0.01, 0.05, -0.02, 0.003],
     np.array([[0.99, 0.01, 0.02], [-0.01, 0.999, 0.03], [-0.02, -0.03, 0.998]]),
     np.array([[600., 0.1, 300], [0, 610, 200], [0, 0, 1]]), (640, 400)),
    (K_DRV, [], None, None, (50, 40)),
]


@pytest.mark.parametrize("K,dist,R,newK,size", CASES)
def test_map_builder_matches_oracle(K, dist, R, newK, size):
    a = S.init_undistort_rectify_map(K, dist, size, R, newK)
    b = O.init_undistort_rectify_map(K, dist, size, R, newK)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert int(a[1].max()) < 1024                      # (fy << 5) | fx


def test_map_builder_properties():
    # no distortion, newK == K: the identity map with zero fractions
    m1, m2 = S.init_undistort_rectify_map(K_DRV, [0, 0, 0, 0], (96, 64))
    xs, ys = np.meshgrid(np.arange(96), np.arange(64))
    assert np.array_equal(m1[..., 0], xs) and np.array_equal(m1[..., 1], ys) and not m2.any()
    # pure radial distortion keeps the principal point: the pixel nearest to it maps (almost) to itself
    K = np.array([[200.0, 0, 40.0], [0, 200.0, 30.0], [0, 0, 1]])
    m1, m2 = S.init_undistort_rectify_map(K, [-0.3, 0.1, 0, 0], (80, 60))
    assert tuple(m1[30, 40]) == (40, 30) and m2[30, 40] == 0
    # barrel distortion (k1 < 0) samples nearer to the centre than the pixel itself, symmetrically
    fix = m1[..., 0].astype(np.int64) * 32 + (m2 & 31)               # x coordinate in 1/32 pixel
    assert fix[30, 79] < 79 * 32 and fix[30, 1] > 1 * 32
    assert abs((fix[30, 79] - 40 * 32) - (40 * 32 - fix[30, 1])) <= 1
    # a new camera matrix with twice the focal length zooms in: the corners sample inside the frame
    newK = K.copy(); newK[0, 0] *= 2; newK[1, 1] *= 2
    z1, _ = S.init_undistort_rectify_map(K, [0, 0, 0, 0], (80, 60), newK=newK)
    assert tuple(z1[0, 0]) == (20, 15)
    with pytest.raises(capi.StitchError):
        S.init_undistort_rectify_map(K, [0.1, 0.2, 0.3], (8, 8))          # 3 coefficients: not a shape OpenCV takes


def test_lens_map_composition_matches_oracle_and_identity():
    """MST_FLAG_LENS_COMPOSED (SURVEY 8 f1 as written): the host composition agrees with the oracle bit for bit; undistorted
    lens maps (the identity) leave the warp's map unchanged inside the frame."""
    rng = np.random.default_rng(11)
    w, h = 333, 211
    K = np.array([[w * 0.6, 0, w * 0.47], [0, w * 0.62, h * 0.55], [0, 0, 1]])
    m1, m2 = S.init_undistort_rectify_map(K, [0.1, -0.05, 0.001, 0.002, 0.03], (w, h))
    x = rng.uniform(-20, w + 20, 5000).astype(np.float32)
    y = rng.uniform(-20, h + 20, 5000).astype(np.float32)
    x[:7] = [np.nan, -1.0, 0.0, w - 1, w - 1.0001, np.inf, -np.inf]
    y[:7] = [3.0, -1.0, 0.0, h - 1, h - 1.0001, 5.0, 5.0]
    a = S.compose_lens_maps(x, y, m1, m2)
    b = O.compose_lens_maps(x, y, m1, m2)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    i1, i2 = S.init_undistort_rectify_map(K, [0, 0, 0, 0], (w, h))
    inside = (x >= 0) & (x <= w - 1) & (y >= 0) & (y <= h - 1)
    cx, cy = S.compose_lens_maps(x, y, i1, i2)
    assert np.abs(cx[inside] - x[inside]).max() < 1e-4 and np.abs(cy[inside] - y[inside]).max() < 1e-4
    # a one-row lens map: the row alone is used
    r1, r2 = S.init_undistort_rectify_map(np.array([[50.0, 0, 20], [0, 50, 0], [0, 0, 1]]), [0.05, 0, 0, 0], (40, 1))
    a = S.compose_lens_maps(x[:200], y[:200], r1, r2)
    b = O.compose_lens_maps(x[:200], y[:200], r1, r2)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def _frame(w, h, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3)).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("size,dist", [((1280, 720), D_DRV), ((1280, 720), D_YML), ((160, 120), [-0.9, 0.6, 0.02, -0.03]), ((31, 17), [0.4, 0, 0, 0])])
def test_remap_fixed_bit_exact(size, dist):
    w, h = size
    K = K_YML if dist is D_YML else K_DRV if size == (1280, 720) else np.array([[w * 0.6, 0, w * 0.47], [0, w * 0.62, h * 0.55], [0, 0, 1]])
    m1, m2 = S.init_undistort_rectify_map(K, dist, size)
    img = _frame(w, h, 5)
    assert np.array_equal(S.remap_fixed(img, m1, m2), O.remap_fixed(img, m1, m2))
    # arbitrary (not model-generated) maps: far outside taps, every fraction
    rng = np.random.default_rng(9)
    r1 = rng.integers(-40, max(w, h) + 40, (h, w, 2)).astype(np.int16)
    r2 = rng.integers(0, 1024, (h, w)).astype(np.uint16)
    assert np.array_equal(S.remap_fixed(img, r1, r2), O.remap_fixed(img, r1, r2))


@pytest.mark.gpu
@pytest.mark.parametrize("blend", ["multiband", "feather", "no"])
def test_compose_with_undistort_prepass(blend):
    """The shipping loop (DRV:741,752): undistort every frame, then warp / compensate / blend."""
    import torch
    cfg = rigmod.scaled_config("C3", 8)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    w, h = rig["size"]
    K = np.array([[w * 0.52, 0, w * 0.49], [0, w * 0.52, h * 0.51], [0, 0, 1]])
    maps = []
    for i, c in enumerate(rig["cameras"]):
        if i == 4:
            maps.append(None)                       # a camera without lens maps keeps its frames
            continue
        maps.append(S.init_undistort_rectify_map(K, [-0.35 + 0.02 * i, 0.15, 0.001 * i, -0.002], (w, h)))
        c["undistort"] = maps[-1]
    eng = H.make_engine(rig, max_batch=3, overlap_frames=(blend == "multiband"))
    _, _, cw, ch = eng.canvas()
    host = [rigmod.frames_for(rig, i) for i in range(3)]
    expect = []
    for fr in host:
        und = [f if m is None else O.remap_fixed(f, m[0], m[1]) for f, m in zip(fr, maps)]
        expect.append(H.oracle_compose(rig, und))
    pano, mask = eng.compose(host[0])                                    # host path
    assert np.array_equal(pano, expect[0][1]) and np.array_equal(mask, expect[0][2])
    dev = torch.device("cuda", 0)
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
    ptrs = [[t.data_ptr() for t in fr] for fr in dfr]
    strides = [[t.stride(0) for t in fr] for fr in dfr]
    torch.cuda.synchronize()
    eng.compose_device_batch(ptrs, strides, [p.data_ptr() for p in panos], panos[0].stride(0))      # batch of 3
    eng.synchronize()
    for i in range(3):
        assert np.array_equal(panos[i].cpu().numpy(), expect[i][1]), "frame set %d" % i
    for rep in range(2):                                                 # pipelined single frames re-use both sets
        for i in range(3):
            panos[i].zero_()
            eng.compose_device(ptrs[i], strides[i], panos[i].data_ptr(), panos[i].stride(0))
    eng.synchronize()
    for i in range(3):
        assert np.array_equal(panos[i].cpu().numpy(), expect[i][1]), "pipelined frame %d" % i
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("blend,warp", [("multiband", "spherical"), ("feather", "plane"), ("no", "cylindrical"), ("multiband", "fisheye")])
def test_compose_with_lens_maps_composed_into_the_warp(blend, warp):
    """SURVEY 8 f1 as written (MST_FLAG_LENS_COMPOSED): one resampling of the RAW frames through the composed map, no
    pre-pass; parity against the oracle's statement of the same composition, single frames and a batch."""
    import torch
    cfg = rigmod.scaled_config("C3" if warp != "plane" else "C2", 8)
    cfg["blend"] = blend
    cfg["warp"] = warp
    rig = rigmod.build_rig(cfg)
    w, h = rig["size"]
    K = np.array([[w * 0.52, 0, w * 0.49], [0, w * 0.52, h * 0.51], [0, 0, 1]])
    maps = []
    for i, c in enumerate(rig["cameras"]):
        if i == 1:
            maps.append(None)                       # a camera without lens maps: the plain map, read from memory
            continue
        maps.append(S.init_undistort_rectify_map(K, [-0.35 + 0.02 * i, 0.15, 0.001 * i, -0.002], (w, h)))
        c["undistort"] = maps[-1]
    eng = H.make_engine(rig, max_batch=3, lens_composed=True)
    _, _, cw, ch = eng.canvas()
    host = [rigmod.frames_for(rig, i) for i in range(3)]

    def expect(frames):
        cams = []
        for c, f, m in zip(rig["cameras"], frames, maps):
            cams.append(dict(K=c["K"], R=c["R"], frame=f, mask=c["mask"], gain_map=c.get("gain_map"), gain=c.get("gain", 1.0),
                             lens_maps=m))
        return O.compose_panorama(H.WARP_O[rig["warp"]], rig["scale"], H.BLEND_O[blend], H.EXPOS_O[rig["expos"]], cams)

    exp = [expect(fr) for fr in host]
    pano, mask = eng.compose(host[0])
    assert np.array_equal(pano, exp[0][1]), H.diff_report(pano, exp[0][1])
    assert np.array_equal(mask, exp[0][2])
    # (and it is NOT the two-stage result: the pre-pass variant differs wherever the lens bends)
    two = H.oracle_compose(rig, [f if m is None else O.remap_fixed(f, m[0], m[1]) for f, m in zip(host[0], maps)])
    assert not np.array_equal(two[1], exp[0][1])
    dev = torch.device("cuda", 0)
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    eng.compose_device_batch([[t.data_ptr() for t in fr] for fr in dfr], [[t.stride(0) for t in fr] for fr in dfr],
                             [p.data_ptr() for p in panos], panos[0].stride(0))
    eng.synchronize()
    for i in range(3):
        assert np.array_equal(panos[i].cpu().numpy(), exp[i][1]), "frame set %d" % i
    eng.close()

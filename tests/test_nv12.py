"""NV12 ingest (optional): cvtColor(COLOR_YUV2BGR_NV12) of OpenCV 2.4.11 restated in the oracle (known
answers by hand), the device conversion against it, and the compose calls of an engine whose frames are NV12."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import capi, rig as rigmod


def _nv12(w, h, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (h * 3 // 2, w)).astype(np.uint8)
    f[0, :8] = [0, 15, 16, 17, 128, 234, 235, 255]          # luma below / at / above the nominal range
    f[h, :8] = [0, 255, 255, 0, 128, 128, 16, 240]           # chroma extremes (U V pairs)
    return f


def test_oracle_known_answers():
    # grey: U = V = 128 -> B = G = R = ((Y - 16) * 1220542 + 2^19) >> 20: 16 -> 0, 235 -> 255, 126 -> 128
    f = np.full((6, 4), 128, np.uint8)
    f[:4] = np.array([[16, 235, 126, 0]] * 4, np.uint8)
    out = O.cvt_nv12_bgr(f)
    assert out.shape == (4, 4, 3)
    assert [tuple(p) for p in out[0]] == [(0, 0, 0), (255, 255, 255), (128, 128, 128), (0, 0, 0)]
    # saturated red-difference: V = 255, U = 128, Y = 81 -> R = (65*1220542 + 2^19 + 1673527*127) >> 20 = 278 -> 255,
    # G = (65*1220542 + 2^19 - 852492*127) >> 20 = -27.1.. -> 0 (arithmetic shift of a negative sum, then saturation)
    f = np.zeros((3, 2), np.uint8)
    f[:2] = 81
    f[2] = [128, 255]
    out = O.cvt_nv12_bgr(f)
    assert tuple(out[0, 0]) == (((65 * 1220542 + (1 << 19)) >> 20), 0, 255)
    # one chroma pair serves a 2 x 2 block of luma
    f = np.zeros((6, 4), np.uint8)
    f[:4] = 100
    f[4] = [90, 200, 200, 60]
    out = O.cvt_nv12_bgr(f)
    assert (out[0:2, 0:2] == out[0, 0]).all() and (out[0:2, 2:4] == out[0, 2]).all() and not (out[0, 0] == out[0, 2]).all()


def test_engine_rejects_odd_frames_without_a_gpu_too():
    L = capi.load()
    assert L.mst_cvt_nv12_bgr(0, None, 4, 4, 4, None, 12) == capi.ERR_INVALID_ARG
    buf = np.zeros((9, 6), np.uint8)
    out = np.zeros((6, 6, 3), np.uint8)
    assert L.mst_cvt_nv12_bgr(0, buf.ctypes.data, 5, 6, 6, out.ctypes.data, 18) == capi.ERR_INVALID_ARG      # odd width


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1920, 1080), (64, 36), (6, 2), (130, 10)])
def test_device_conversion_bit_exact(size):
    w, h = size
    f = _nv12(w, h, 3) if w >= 8 else np.random.default_rng(1).integers(0, 256, (h * 3 // 2, w)).astype(np.uint8)
    assert np.array_equal(S.cvt_nv12_bgr(f), O.cvt_nv12_bgr(f))
    pad = np.zeros((f.shape[0], w + 5), np.uint8)              # a row stride that is not the width
    pad[:, :w] = f
    assert np.array_equal(S.cvt_nv12_bgr(pad[:, :w]), O.cvt_nv12_bgr(f))


@pytest.mark.gpu
@pytest.mark.parametrize("blend", ["multiband", "feather", "no"])
def test_compose_from_nv12_frames(blend):
    import torch
    cfg = rigmod.scaled_config("C3", 8)
    cfg["blend"] = blend
    cfg["size"] = (cfg["size"][0], cfg["size"][1] + 1)           # 240 x 136: NV12 frames have even sizes
    rig = rigmod.build_rig(cfg)
    w, h = rig["size"]
    assert w % 2 == 0 and h % 2 == 0
    n = len(rig["cameras"])
    sets = [[_nv12(w, h, 100 + 10 * t + i) for i in range(n)] for t in range(3)]
    exp = [H.oracle_compose(rig, [O.cvt_nv12_bgr(f) for f in fr]) for fr in sets]
    eng = H.make_engine(rig, max_batch=3, frames_nv12=True)
    _, _, cw, ch = eng.canvas()
    pano, mask = eng.compose(sets[0])                                    # host frames: half the bytes over PCIe
    assert np.array_equal(pano, exp[0][1]), H.diff_report(pano, exp[0][1])
    assert np.array_equal(mask, exp[0][2])
    dev = torch.device("cuda", 0)
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in sets]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    eng.compose_device_batch([[t.data_ptr() for t in fr] for fr in dfr], [[t.stride(0) for t in fr] for fr in dfr],
                             [p.data_ptr() for p in panos], panos[0].stride(0))
    eng.synchronize()
    for i in range(3):
        assert np.array_equal(panos[i].cpu().numpy(), exp[i][1]), "frame set %d" % i
    # the three-stage host pipeline
    out = [np.empty((ch, cw, 3), np.uint8) for _ in range(3)]
    for i in range(3):
        eng.pipeline_submit(sets[i], out[i])
    for i in range(3):
        eng.pipeline_wait()
        assert np.array_equal(out[i], exp[i][1]), "pipelined frame %d" % i
    with pytest.raises(capi.StitchError):
        eng.stage_cameras([0], {0: dfr[0][0].data_ptr()}, {0: dfr[0][0].stride(0)})
    eng.close()
    # calls in flight (mst_config.overlap_depth): every call converts into its own slot's BGR frames; with lens maps on two
    # cameras the undistort pre-pass runs per slot as well.  Nine one-frame-set calls, three panorama buffers in rotation.
    K = np.array([[0.55 * w, 0, 0.5 * w], [0, 0.55 * w, 0.5 * h], [0, 0, 1]])
    und = S.init_undistort_rectify_map(K, [-0.3, 0.12, 0.002, -0.001], (w, h))
    for c in rig["cameras"][:2]:
        c["undistort"] = und
    exp_u = []
    for fr in sets:
        bgr = [O.cvt_nv12_bgr(f) for f in fr]
        bgr = [O.remap_fixed(b, und[0], und[1]) if i < 2 else b for i, b in enumerate(bgr)]
        exp_u.append(H.oracle_compose(rig, bgr))
    eng = H.make_engine(rig, frames_nv12=True, overlap_frames=True, overlap_depth=3)
    for k in range(9):
        i = k % 3
        eng.compose_device([t.data_ptr() for t in dfr[i]], [t.stride(0) for t in dfr[i]], panos[i].data_ptr(), panos[i].stride(0))
    eng.synchronize()
    for i in range(3):
        assert np.array_equal(panos[i].cpu().numpy(), exp_u[i][1]), "in flight, frame set %d: %s" % (i, H.diff_report(panos[i].cpu().numpy(), exp_u[i][1]))
    eng.close()

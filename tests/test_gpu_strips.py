"""Canvas split into column strips (mst_config.strip_index / strip_count, SURVEY.md 8e second form), emulated on one
GPU: G engines, engine k composing strip k, each into a strip-sized buffer.  Every strip must equal the same columns of
the oracle's panorama bit for bit (the restricted work lists compute exactly what the strip depends on), the strips
must tile the canvas, and the host-side strip arithmetic must agree with the engine's."""
import numpy as np
import pytest

import helpers as H
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod, sharding as sh

pytestmark = pytest.mark.gpu


def _compose_strips(rig, frames, blend, count, bands=5, lut=False, batch=1):
    import torch
    dev = torch.device("cuda", 0)
    n = len(rig["cameras"])
    d_frames = [torch.from_numpy(f).to(dev) for f in frames]
    ptrs, strides = [t.data_ptr() for t in d_frames], [t.stride(0) for t in d_frames]
    strips, canvas = [], None
    for k in range(count):
        eng = S.PanoramaEngine(n, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=bands, driver_lut=lut,
                               strip=(k, count), max_batch=batch)
        eng.update(rig["scale"], rig["cameras"])
        (cx, cy, cw, ch), _ = eng.acquire()
        sx, sy, sw, shh = eng.strip()
        assert (sy, shh) == (cy, ch) and cx <= sx and sx + sw <= cx + cw
        canvas = (cx, cy, cw, ch)
        buf = torch.full((ch, max(sw, 1), 3), 77, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()         # fills run on torch's stream; the engine's stream does not wait for it
        if sw > 0:
            if batch == 1:
                eng.compose_device(ptrs, strides, buf.data_ptr() - (sx - cx) * 3, buf.stride(0))
            else:
                bufs = [torch.full((ch, sw, 3), 77, dtype=torch.uint8, device=dev) for _ in range(batch)]
                torch.cuda.synchronize()
                eng.compose_device_batch([ptrs] * batch, [strides] * batch, [b.data_ptr() - (sx - cx) * 3 for b in bufs], bufs[0].stride(0))
                torch.cuda.synchronize()
                assert all(torch.equal(b, bufs[0]) for b in bufs)
                buf = bufs[0]
            torch.cuda.synchronize()
        strips.append((sx - cx, sw, buf.cpu().numpy()))
        eng.close()
    return canvas, strips


@pytest.mark.parametrize("name,div,blend,count,bands", [
    ("C3", 4, "multiband", 2, 5), ("C3", 4, "multiband", 4, 5), ("C3", 4, "multiband", 3, 3),
    ("C2", 2, "feather", 2, 5), ("C3", 4, "no", 4, 5), ("C5", 8, "multiband", 8, 4)])
def test_strips_tile_the_oracle_panorama(name, div, blend, count, bands):
    cfg = rigmod.scaled_config(name, div)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    frames = rigmod.frames_for(rig, 2)
    rect, opano, _ = H.oracle_compose(rig, frames, blend=blend, num_bands=bands)
    canvas, strips = _compose_strips(rig, frames, blend, count, bands)
    assert canvas == rect
    cw = rect[2]
    assert strips[0][0] == 0 and strips[-1][0] + strips[-1][1] == cw
    for (x, w, got), nxt in zip(strips, strips[1:] + [(cw, 0, None)]):
        assert x + w == nxt[0]                                               # the strips tile the canvas
        if w:
            assert np.array_equal(got[:, :w], opano[:, x:x + w]), (x, w, H.diff_report(got[:, :w], opano[:, x:x + w]))
    # the host-side arithmetic (what every rank computes for its peers) agrees with the engines
    width = cw if blend != "multiband" else (cw + (1 << bands) - 1) // (1 << bands) * (1 << bands)
    for k, (x, w, _) in enumerate(strips):
        x0, x1 = sh.strip_columns(width, k, count)
        assert (min(x0, cw), min(x1, cw) - min(x0, cw)) == (x, w)


def test_strips_full_size_c4_batched():
    """BASELINE C4 at full size (8 x 1080p, canvas 11308 x 1050) cut into four strips, three frame sets per launch
    (the staged warp kernel's regime)."""
    rig = rigmod.build_rig(rigmod.CONFIGS["C4"])
    frames = rigmod.frames_for(rig, 1)
    rect, opano, _ = H.oracle_compose(rig, frames, threads=32)
    canvas, strips = _compose_strips(rig, frames, "multiband", 4, batch=3)
    assert canvas == rect
    for x, w, got in strips:
        assert np.array_equal(got[:, :w], opano[:, x:x + w]), (x, w)


def test_strip_of_the_driver_lut():
    cfg = rigmod.scaled_config("C3", 4)
    cfg["blend"] = "no"
    rig = rigmod.build_rig(cfg)
    frames = rigmod.frames_for(rig, 0)
    n = len(rig["cameras"])
    whole = S.PanoramaEngine(n, warp=rig["warp"], blend="no", expos=rig["expos"], driver_lut=True)
    whole.update(rig["scale"], rig["cameras"])
    ref, _ = whole.compose(frames)
    whole.close()
    canvas, strips = _compose_strips(rig, frames, "no", 3, lut=True)
    for x, w, got in strips:
        assert np.array_equal(got[:, :w], ref[:, x:x + w]), (x, w)


def _compose_strips_with_halo(rig, frames_sets, count, bands=5):
    """G strip engines with MST_FLAG_STRIP_HALO on one GPU: every front half, the exchange (a neighbour's send buffer IS the
    other's receive buffer), every back half.  Returns the canvas, [(x, w, [strip of set 0, ...])] and the halo bytes."""
    import torch
    dev = torch.device("cuda", 0)
    n = len(rig["cameras"])
    batch = len(frames_sets)
    d_sets = [[torch.from_numpy(f).to(dev) for f in frames] for frames in frames_sets]
    ptrs = [[t.data_ptr() for t in fs] for fs in d_sets]
    strides = [[t.stride(0) for t in fs] for fs in d_sets]
    engines, geo, send = [], [], []
    for k in range(count):
        eng = S.PanoramaEngine(n, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=bands, strip=(k, count),
                               strip_halo=True, max_batch=batch)
        eng.update(rig["scale"], rig["cameras"])
        (cx, cy, cw, ch), _ = eng.acquire()
        sx, _, sw, _ = eng.strip()
        engines.append(eng)
        geo.append((cx, cy, cw, ch, sx, sw))
    halo = [e.halo_bytes() for e in engines]
    for k in range(count):               # both sides derive the same rectangles from the same calibration
        if k + 1 < count:
            assert halo[k][0][1] == halo[k + 1][1][0] and halo[k][1][1] == halo[k + 1][0][0], (k, halo)
    assert halo[0][0][0] == 0 and halo[0][1][0] == 0 and halo[-1][0][1] == 0 and halo[-1][1][1] == 0
    for k, eng in enumerate(engines):
        bufs = [torch.full((max(1, batch * b),), 0xA5, dtype=torch.uint8, device=dev) for b in halo[k][0]]
        torch.cuda.synchronize()
        eng.strip_front_batch(ptrs, strides, bufs[0].data_ptr() if halo[k][0][0] else None, bufs[1].data_ptr() if halo[k][0][1] else None)
        send.append(bufs)
    for eng in engines:
        eng.synchronize()
    out = []
    for k, eng in enumerate(engines):
        cx, cy, cw, ch, sx, sw = geo[k]
        panos = [torch.full((ch, max(sw, 1), 3), 77, dtype=torch.uint8, device=dev) for _ in range(batch)]
        torch.cuda.synchronize()
        left = send[k - 1][1].data_ptr() if k > 0 and halo[k][1][0] else None
        right = send[k + 1][0].data_ptr() if k + 1 < count and halo[k][1][1] else None
        eng.strip_back_batch(batch, [p.data_ptr() - (sx - cx) * 3 for p in panos], panos[0].stride(0), left, right)
        eng.synchronize()
        out.append((sx - cx, sw, [p.cpu().numpy() for p in panos]))
    for eng in engines:
        eng.close()
    return geo[0][:4], out, halo


@pytest.mark.parametrize("name,div,count,bands,batch", [("C3", 4, 2, 5, 1), ("C3", 4, 3, 5, 3), ("C3", 2, 4, 5, 2), ("C5", 4, 4, 4, 1),
                                                        ("C3", 4, 2, 3, 1), ("C3", 4, 2, 2, 1)])
def test_strips_with_halo_exchange_tile_the_oracle_panorama(name, div, count, bands, batch):
    """MST_FLAG_STRIP_HALO: neighbouring strips exchange the halo of the cameras' Gaussian level 2 instead of recomputing it.
    Every strip of every frame set of the batch must still equal the oracle's panorama columns bit for bit; interior
    boundaries carry a payload in both directions (bands = 2: the pyramid is too short to cut, nothing is exchanged)."""
    cfg = rigmod.scaled_config(name, div)
    cfg["blend"] = "multiband"
    rig = rigmod.build_rig(cfg)
    sets = [rigmod.frames_for(rig, 5 + s) for s in range(batch)]
    want = [H.oracle_compose(rig, frames, blend="multiband", num_bands=bands, threads=16) for frames in sets]
    canvas, strips, halo = _compose_strips_with_halo(rig, sets, count, bands)
    assert canvas == want[0][0]
    if bands >= 3:
        assert all(halo[k][0][1] > 0 and halo[k + 1][0][0] > 0 for k in range(count - 1)), halo
    else:
        assert all(b == 0 for h in halo for pair in h for b in pair), halo
    for x, w, got in strips:
        for s in range(batch):
            opano = want[s][1]
            assert np.array_equal(got[s][:, :w], opano[:, x:x + w]), (x, w, s, H.diff_report(got[s][:, :w], opano[:, x:x + w]))


def test_halo_strip_computes_less_than_a_self_sufficient_strip():
    """What the exchange removes: level-0 pixels a strip needs (its pyramids stop at its own columns below the cut).  The
    neighbours' rectangles are bounding boxes, so the saving is smaller than the halo's width suggests."""
    cfg = rigmod.scaled_config("C3", 2)
    rig = rigmod.build_rig(cfg)
    n = len(rig["cameras"])
    tiles = {}
    for halo in (False, True):
        eng = S.PanoramaEngine(n, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, strip=(1, 4), strip_halo=halo)
        eng.update(rig["scale"], rig["cameras"])
        tiles[halo] = sum(int(eng.debug_read(S.capi.DBG_CAM_NEED_U8, c, 0).sum()) for c in range(n))
        eng.close()
    print("level-0 pixels needed: self-sufficient %d, with halo exchange %d" % (tiles[False], tiles[True]))
    assert tiles[True] < tiles[False], tiles

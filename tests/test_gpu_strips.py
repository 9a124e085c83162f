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

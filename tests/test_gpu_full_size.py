"""Every BASELINE config at full size in the driver-run suite (SURVEY.md 8d geometry), one frame each, engine vs
oracle, bit for bit: C2 (2x1080p plane + feather), C3 (test_gpu_parity.py), C4 (8x1080p spherical, through the
camera-sharded stage calls of two engines -- the one-GPU emulation of the per-GPU split) and C5 (6x4K cylindrical,
5 bands: the config closest to the 32-bit offset / 24-bit row-product / map-table window limits)."""
import os

import numpy as np
import pytest

import helpers as H
from stitchingvideo_amd import rig as rigmod

pytestmark = pytest.mark.gpu

THREADS = min(32, os.cpu_count() or 1)


def test_full_size_c2_feather():
    rig = rigmod.build_rig(rigmod.CONFIGS["C2"])
    frames = rigmod.frames_for(rig, 1)
    eng = H.make_engine(rig)
    assert eng.canvas()[2:] == (3259, 1406)
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, threads=THREADS)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)
    eng.close()


def test_full_size_c5_4k_cylindrical_multiband():
    rig = rigmod.build_rig(rigmod.CONFIGS["C5"])
    frames = rigmod.frames_for(rig, 1)
    eng = H.make_engine(rig)
    assert eng.canvas()[2:] == (15706, 2159)
    assert eng.num_bands() == 5
    pano, mask = eng.compose(frames)
    rect, opano, omask = H.oracle_compose(rig, frames, threads=THREADS)
    assert eng.canvas() == rect
    assert np.array_equal(mask, omask), H.diff_report(mask, omask)
    assert np.array_equal(pano, opano), H.diff_report(pano, opano)
    eng.close()


def test_full_size_c4_camera_sharded_two_engines():
    """C4 as specified is 8 cameras on 8 GPUs with one gather; on one GPU two engines stand in for two ranks:
    each stages its own cameras, the hand-off payload of rank 1's cameras is copied into rank 0 (what the RCCL
    point-to-point exchange does), rank 0 runs the canvas stage.  Also checked: the plain one-engine compose."""
    import torch
    from stitchingvideo_amd import sharding as sh
    rig = rigmod.build_rig(rigmod.CONFIGS["C4"])
    frames = rigmod.frames_for(rig, 2)
    dev = torch.device("cuda", 0)
    e0, e1 = H.make_engine(rig), H.make_engine(rig)
    assert e0.canvas()[2:] == (11308, 1050)
    n = len(frames)
    dfr = [torch.from_numpy(f).to(dev) for f in frames]
    ptrs = {i: dfr[i].data_ptr() for i in range(n)}
    strides = {i: dfr[i].stride(0) for i in range(n)}
    mine0, mine1 = sh.streams_of_rank(0, 2, n), sh.streams_of_rank(1, 2, n)
    _, _, cw, ch = e0.canvas()
    pano = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        e0.stage_cameras(mine0, {i: ptrs[i] for i in mine0}, {i: strides[i] for i in mine0}, stream=stream.cuda_stream)
        e1.stage_cameras(mine1, {i: ptrs[i] for i in mine1}, {i: strides[i] for i in mine1}, stream=stream.cuda_stream)
        moved = sh.copy_payload(e1, e0, mine1, dev, stream=stream.cuda_stream)
        e0.stage_canvas(pano.data_ptr(), pano.stride(0), stream=stream.cuda_stream)
    stream.synchronize()
    # the hand-off is the launched tile interiors only: every camera -- the wrap-around one (roi 11308 x 1050, pyramid
    # block 52 MB: level 0 packed BGR8, the levels above a dword a pixel) included -- ships a payload of a 1080p camera's size
    sizes = [e0.payload_bytes(c) for c in range(n)]
    blocks = [e0.pyramid_buffer(c)[1] for c in range(n)]
    assert max(blocks) > 45e6 and max(sizes) <= 20e6, (sizes, blocks)
    assert moved == sum(sizes[c] for c in mine1)
    rect, opano, omask = H.oracle_compose(rig, frames, threads=THREADS)
    assert e0.canvas() == rect
    assert np.array_equal(pano.cpu().numpy(), opano), H.diff_report(pano.cpu().numpy(), opano)
    p1, m1 = e1.compose(frames)
    assert np.array_equal(p1, opano) and np.array_equal(m1, omask)
    e0.close(); e1.close()


@pytest.mark.parametrize("in_flight", [1, 4])
def test_full_size_c3_in_the_bench_configuration(in_flight):
    """C3 exactly as bench.py times it: an engine created with max_batch = 8 (which also selects the taller canvas
    tiles of the batched mode) composing eight frame sets in one launch sequence -- and, in_flight = 4, with
    mst_config.overlap_depth = 4 and four such launch sequences in flight, each writing its own eight panoramas (the
    default of bench.py since round 4) -- every panorama against the oracle (two distinct frame sets, alternated)."""
    import torch
    rig = rigmod.build_rig(rigmod.CONFIGS["C3"])
    eng = H.make_engine(rig, max_batch=8, overlap_frames=in_flight > 1, overlap_depth=in_flight if in_flight > 1 else 0)
    _, _, cw, ch = eng.canvas()
    assert (cw, ch) == (7852, 1020)
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(2)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    orders = [[0, 1, 1, 0, 1, 0, 0, 1], [1, 1, 0, 0, 1, 0, 1, 0], [0, 0, 0, 1, 1, 1, 0, 1], [1, 0, 1, 0, 0, 1, 1, 0], [0, 1, 0, 1, 1, 0, 0, 0]][:in_flight + (in_flight > 1)]
    panos = [[torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in order] for order in orders]
    masks = [[torch.zeros((ch, cw), dtype=torch.uint8, device=dev) for _ in order] for order in orders]
    torch.cuda.synchronize()          # the fills run on torch's stream, the engine's stream does not wait for it
    for order, pn, mk in zip(orders, panos, masks):          # (five calls on four slots: the fifth waits for the first's buffers)
        eng.compose_device_batch([[t.data_ptr() for t in dfr[k]] for k in order], [[t.stride(0) for t in dfr[k]] for k in order],
                                 [p.data_ptr() for p in pn], pn[0].stride(0), [m.data_ptr() for m in mk], mk[0].stride(0))
    eng.synchronize()
    expect = [H.oracle_compose(rig, fr, threads=THREADS) for fr in host]
    for c, order in enumerate(orders):
        for i, k in enumerate(order):
            assert np.array_equal(masks[c][i].cpu().numpy(), expect[k][2]), "call %d frame set %d mask" % (c, i)
            assert np.array_equal(panos[c][i].cpu().numpy(), expect[k][1]), "call %d frame set %d: %s" % (c, i, H.diff_report(panos[c][i].cpu().numpy(), expect[k][1]))
    eng.close()

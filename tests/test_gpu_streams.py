"""Device-resident frame loop (mst_compose_device): plain stream order and the frame-pipelined
throughput mode must give the same panoramas, in order, as the oracle."""
import numpy as np
import pytest

import helpers as H
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap", [False, True, 4])
def test_device_frame_loop_matches_oracle(overlap):
    """overlap = 4: mst_config.overlap_depth -- whole calls on four engine-owned streams.  Six panorama buffers in rotation are
    not a multiple of four, so calls on different streams meet on a buffer: the engine orders those."""
    import torch
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    eng = H.make_engine(rig, overlap_frames=bool(overlap), overlap_depth=overlap if overlap is not True and overlap else 0)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    n_frames = 6
    host = [rigmod.frames_for(rig, i) for i in range(n_frames)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(n_frames)]
    masks = [torch.empty((ch, cw), dtype=torch.uint8, device=dev) for _ in range(n_frames)]
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for rep in range(2):               # second pass re-uses both pyramid sets and all events
            for i in range(n_frames):
                eng.compose_device([t.data_ptr() for t in dfr[i]], [t.stride(0) for t in dfr[i]],
                                   panos[i].data_ptr(), panos[i].stride(0), masks[i].data_ptr(), masks[i].stride(0),
                                   stream=stream.cuda_stream)
    stream.synchronize()
    for i in range(n_frames):
        _, opano, omask = H.oracle_compose(rig, host[i])
        assert np.array_equal(panos[i].cpu().numpy(), opano), "frame %d" % i
        assert np.array_equal(masks[i].cpu().numpy(), omask), "frame %d mask" % i
    eng.close()


@pytest.mark.parametrize("depth", [0, 3, 8])
def test_overlap_mode_same_output_buffer_keeps_frame_order(depth):
    """The canvas kernels stay ordered on the caller's stream (two-stream form), calls that write the same buffer are ordered
    behind each other (deep form): re-using one panorama buffer is safe."""
    import torch
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    eng = H.make_engine(rig, overlap_frames=True, overlap_depth=depth)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(5)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    pano = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    for i in range(5):
        eng.compose_device([t.data_ptr() for t in dfr[i]], [t.stride(0) for t in dfr[i]], pano.data_ptr(),
                           pano.stride(0), stream=stream.cuda_stream)
    stream.synchronize()
    _, opano, _ = H.oracle_compose(rig, host[4])
    assert np.array_equal(pano.cpu().numpy(), opano)
    eng.close()


def test_camera_sharded_stages_match_oracle():
    """Two engines stand in for two ranks on one GPU: each runs the per-camera stage for its cameras, the
    odd cameras' pyramid blocks are copied into the 'root' engine (what the RCCL gather does), and the root's
    canvas stage must reproduce the oracle bit for bit."""
    import torch
    from stitchingvideo_amd import sharding as sh
    rig = rigmod.build_rig(rigmod.scaled_config("C4", 8))
    frames = rigmod.frames_for(rig, 2)
    dev = torch.device("cuda", 0)
    e0, e1 = H.make_engine(rig), H.make_engine(rig)
    n = len(frames)
    dfr = [torch.from_numpy(f).to(dev) for f in frames]
    ptrs = {i: dfr[i].data_ptr() for i in range(n)}
    strides = {i: dfr[i].stride(0) for i in range(n)}
    world = 2
    mine0, mine1 = sh.streams_of_rank(0, world, n), sh.streams_of_rank(1, world, n)
    assert sorted(mine0 + mine1) == list(range(n))
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        e0.stage_cameras(mine0, {i: ptrs[i] for i in mine0}, {i: strides[i] for i in mine0}, stream=stream.cuda_stream)
        e1.stage_cameras(mine1, {i: ptrs[i] for i in mine1}, {i: strides[i] for i in mine1}, stream=stream.cuda_stream)
        moved = sh.copy_payload(e1, e0, mine1, dev, stream=stream.cuda_stream)     # the exchange: packed tile interiors
        whole = sum(e1.pyramid_buffer(c)[1] for c in mine1)
        assert 0 < moved <= whole
        _, _, cw, ch = e0.canvas()
        pano = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
        e0.stage_canvas(pano.data_ptr(), pano.stride(0), stream=stream.cuda_stream)
    stream.synchronize()
    _, opano, _ = H.oracle_compose(rig, frames)
    assert np.array_equal(pano.cpu().numpy(), opano), H.diff_report(pano.cpu().numpy(), opano)
    # world size 1 through the same class
    st = sh.ShardedStitcher(e1, 0, 1, device=dev)
    pano2 = torch.empty_like(pano)
    st.compose(ptrs, strides, pano2.data_ptr(), pano2.stride(0))
    e1.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(pano2.cpu().numpy(), opano)
    e0.close(); e1.close()


def test_camera_sharded_pipeline_alternates_pyramid_sets():
    """submit(t) / finish(t) of ShardedStitcher on one rank: frame t+1's per-camera stage is issued before frame t's
    canvas stage, on the other pyramid set -- the order the multi-rank pipeline uses to overlap the exchange."""
    import torch
    from stitchingvideo_amd import sharding as sh
    rig = rigmod.build_rig(rigmod.scaled_config("C4", 8))
    dev = torch.device("cuda", 0)
    eng = H.make_engine(rig, max_batch=2)
    n = len(rig["cameras"])
    host = [rigmod.frames_for(rig, i) for i in range(4)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    _, _, cw, ch = eng.canvas()
    panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
    st = sh.ShardedStitcher(eng, 0, 1, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        def args(t):
            return ({c: dfr[t][c].data_ptr() for c in range(n)}, {c: dfr[t][c].stride(0) for c in range(n)})
        st.submit(0, *args(0))
        for t in range(1, 4):
            st.submit(t, *args(t))                       # per-camera stage of t (other set) ...
            st.finish(t - 1, panos[t - 1].data_ptr(), panos[t - 1].stride(0))     # ... then the canvas stage of t - 1
        st.finish(3, panos[3].data_ptr(), panos[3].stride(0))
    stream.synchronize()
    for t in range(4):
        _, opano, _ = H.oracle_compose(rig, host[t])
        assert np.array_equal(panos[t].cpu().numpy(), opano), "frame %d" % t
    eng.close()


@pytest.mark.parametrize("blend", ["multiband", "feather", "no"])
def test_unaligned_frames_and_odd_strides(blend):
    """Frame bases at every byte alignment, row strides that are not multiples of four, and an odd-stride
    panorama: the kernels address sources through a dword-aligned base plus byte offsets and use paired
    byte stores only when the panorama rows allow it."""
    import torch
    cfg = rigmod.scaled_config("C3", 8)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    eng = H.make_engine(rig)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = rigmod.frames_for(rig, 3)
    ptrs, strides, keep = [], [], []
    for i, f in enumerate(host):
        h, w, _ = f.shape
        stride = 3 * w + (5, 1, 7, 3, 2, 6)[i % 6]
        off = i % 4
        buf = torch.zeros(off + stride * h + 16, dtype=torch.uint8, device=dev)
        rows = buf[off:off + stride * h].view(h, stride)
        rows[:, :3 * w] = torch.from_numpy(f.reshape(h, 3 * w)).to(dev)
        keep.append(buf)
        ptrs.append(buf.data_ptr() + off)
        strides.append(stride)
    pstride = 3 * cw + 5
    pbuf = torch.zeros(1 + pstride * ch + 16, dtype=torch.uint8, device=dev)
    mstride = cw + 3
    mbuf = torch.zeros(1 + mstride * ch + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.compose_device(ptrs, strides, pbuf.data_ptr() + 1, pstride, mbuf.data_ptr() + 1, mstride)
    eng.synchronize()
    pano = pbuf[1:1 + pstride * ch].view(ch, pstride)[:, :3 * cw].reshape(ch, cw, 3).cpu().numpy()
    mask = mbuf[1:1 + mstride * ch].view(ch, mstride)[:, :cw].cpu().numpy()
    _, opano, omask = H.oracle_compose(rig, host)
    assert np.array_equal(pano, opano)
    assert np.array_equal(mask, omask)
    eng.close()


@pytest.mark.parametrize("blend,count", [("multiband", 4), ("multiband", 3), ("feather", 2)])
def test_batched_launch_matches_oracle(blend, count):
    """mst_compose_device_batch: `count` frame sets through one launch sequence, each against the oracle;
    a second, smaller batch and a single frame afterwards re-use the pyramid sets."""
    import torch
    cfg = rigmod.scaled_config("C3", 8)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    eng = H.make_engine(rig, max_batch=4)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(count)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(count)]
    masks = [torch.zeros((ch, cw), dtype=torch.uint8, device=dev) for _ in range(count)]
    torch.cuda.synchronize()
    ptrs = [[t.data_ptr() for t in fr] for fr in dfr]
    strides = [[t.stride(0) for t in fr] for fr in dfr]
    eng.compose_device_batch(ptrs, strides, [p.data_ptr() for p in panos], panos[0].stride(0),
                             [m.data_ptr() for m in masks], masks[0].stride(0))
    eng.synchronize()
    expect = [H.oracle_compose(rig, host[i]) for i in range(count)]
    for i in range(count):
        assert np.array_equal(panos[i].cpu().numpy(), expect[i][1]), "frame set %d" % i
        assert np.array_equal(masks[i].cpu().numpy(), expect[i][2]), "frame set %d mask" % i
    # reversed order in a batch of two, then a plain single frame
    for p in panos:
        p.zero_()
    eng.compose_device_batch([ptrs[1], ptrs[0]], [strides[1], strides[0]], [panos[0].data_ptr(), panos[1].data_ptr()],
                             panos[0].stride(0))
    eng.compose_device(ptrs[count - 1], strides[count - 1], panos[2 % count].data_ptr(), panos[0].stride(0))
    eng.synchronize()
    assert np.array_equal(panos[0].cpu().numpy(), expect[1][1])
    if count > 2:
        assert np.array_equal(panos[1].cpu().numpy(), expect[0][1])
        assert np.array_equal(panos[2].cpu().numpy(), expect[count - 1][1])
    if blend == "multiband":           # one pyramid set per frame set: max_batch bounds the batch
        with pytest.raises(Exception):
            eng.compose_device_batch([ptrs[0]] * 5, [strides[0]] * 5, [panos[0].data_ptr()] * 5, panos[0].stride(0))
    eng.close()


def test_recalibration_from_a_second_thread_while_composing():
    """The reference recalibrates every 30 s on a worker thread while the frame loop keeps running (DRV:842-847):
    mst_update from another thread must not disturb composes in flight, and takes effect at a later compose."""
    import threading
    import torch
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    cams_b = [dict(c) for c in rig["cameras"]]
    for c in cams_b:
        c["gain_map"] = (c["gain_map"] * np.float32(0.8)).astype(np.float32)
    rig_b = dict(rig, cameras=cams_b)
    eng = H.make_engine(rig, max_batch=2, overlap_frames=True)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = rigmod.frames_for(rig, 7)
    dfr = [torch.from_numpy(f).to(dev) for f in host]
    ptrs, strides = [t.data_ptr() for t in dfr], [t.stride(0) for t in dfr]
    pano = torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev)
    _, pano_a, _ = H.oracle_compose(rig, host)
    _, pano_b, _ = H.oracle_compose(rig_b, host)
    torch.cuda.synchronize()
    errors = []

    def recalibrate():
        try:
            for i in range(6):
                eng.update(rig["scale"], cams_b if i % 2 == 0 else rig["cameras"])
            eng.update(rig["scale"], cams_b)
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=recalibrate)
    t.start()
    seen = set()
    for i in range(300):
        eng.compose_device(ptrs, strides, pano.data_ptr(), pano.stride(0))
        if i % 25 == 0:
            eng.synchronize()
            got = pano.cpu().numpy()
            assert np.array_equal(got, pano_a) or np.array_equal(got, pano_b), "compose %d saw a torn calibration" % i
            seen.add(bool(np.array_equal(got, pano_b)))
    t.join()
    assert not errors, errors
    eng.compose_device(ptrs, strides, pano.data_ptr(), pano.stride(0))
    eng.synchronize()
    assert np.array_equal(pano.cpu().numpy(), pano_b)          # the last update won
    eng.close()


@pytest.mark.parametrize("depth", [0, 4])
def test_recalibration_that_changes_the_canvas_size_while_composing(depth):
    """(depth = 4: the same on an engine with mst_config.overlap_depth = 4 -- device calls run on the engine's own streams.)
    A recalibration that moves R / scale changes the canvas rectangle.  The frame thread latches the calibration
    (mst_acquire_calibration = `if (Upflag) UpdateMat();`, DRV:835-836), sizes its buffers from the latched canvas and
    composes against exactly that state, while a second thread keeps publishing calibrations of two different canvas
    sizes: every panorama must be one of the two oracle panoramas at its own size (host-buffer and device paths)."""
    import threading
    import torch
    cfg_a = rigmod.scaled_config("C3", 8)
    cfg_b = dict(cfg_a, focal=cfg_a["focal"] * 1.13, yaws=[y + 7.0 for y in cfg_a["yaws"]])
    rig_a, rig_b = rigmod.build_rig(cfg_a), rigmod.build_rig(cfg_b)
    host = rigmod.frames_for(rig_a, 3)
    rect_a, pano_a, _ = H.oracle_compose(rig_a, host)
    rect_b, pano_b, _ = H.oracle_compose(rig_b, host)
    assert rect_a[2:] != rect_b[2:] and rect_b[3] > rect_a[3]          # B is wider AND taller than A
    eng = H.make_engine(rig_a, overlap_frames=depth > 0, overlap_depth=depth)
    dev = torch.device("cuda", 0)
    dfr = [torch.from_numpy(f).to(dev) for f in host]
    ptrs, strides = [t.data_ptr() for t in dfr], [t.stride(0) for t in dfr]
    errors, stop = [], threading.Event()

    def recalibrate():
        try:
            i = 0
            while not stop.is_set() and i < 40:
                r = rig_b if i % 2 == 0 else rig_a
                eng.update(r["scale"], r["cameras"])
                i += 1
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=recalibrate)
    t.start()
    seen = set()
    try:
        for i in range(60):
            if i % 2 == 0:                       # host-buffer path: sizes from the latch inside compose()
                pano, mask = eng.compose(host)
            else:                                # device path: explicit latch, then buffers, then compose
                (_, _, cw, ch), gen = eng.acquire()
                d_pano = torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()         # the fill runs on torch's stream; the engine's stream does not wait for it
                eng.compose_device(ptrs, strides, d_pano.data_ptr(), d_pano.stride(0))
                eng.synchronize()
                assert eng.canvas()[2:] == (cw, ch)          # still the latched one, whatever was published meanwhile
                pano = d_pano.cpu().numpy()
            want = pano_a if pano.shape == pano_a.shape else pano_b
            assert pano.shape == want.shape and np.array_equal(pano, want), "compose %d: wrong geometry or pixels" % i
            seen.add(pano.shape == pano_b.shape)
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    eng.update(rig_b["scale"], rig_b["cameras"])
    pano, _ = eng.compose(host)
    assert np.array_equal(pano, pano_b) and eng.canvas() == rect_b
    eng.close()


def test_host_pipeline_three_stage_overlap_matches_oracle():
    """mst_pipeline_submit / mst_pipeline_wait: H2D of t+1, kernels of t, D2H of t-1 on three streams over a ring of three
    slots, page-locked host buffers.  Panoramas come back in submission order, each equal to the oracle's; a
    recalibration in the middle changes the canvas of the frames submitted after it and of no other."""
    rig_a = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    cfg_b = rigmod.scaled_config("C3", 8)
    cfg_b["focal"] *= 1.1
    rig_b = rigmod.build_rig(cfg_b)
    eng = H.make_engine(rig_a)
    n_frames = 9
    host = [rigmod.frames_for(rig_a, i) for i in range(n_frames)]
    pinned_frames = []
    for fr in host:
        row = []
        for f in fr:
            pa = S.PinnedArray(f.shape)
            pa.array[...] = f
            row.append(pa)
        pinned_frames.append(row)
    ra = H.oracle_compose(rig_a, host[0])[0]
    rb = H.oracle_compose(rig_b, host[0])[0]
    big = (max(ra[3], rb[3]), max(ra[2], rb[2]), 3)
    panos = [S.PinnedArray(big) for _ in range(3)]
    results = []

    def wait_one():
        addr, rect = eng.pipeline_wait()
        slot = [i for i, p in enumerate(panos) if p.array.ctypes.data == addr][0]
        w, h = rect[2], rect[3]
        flat = panos[slot].array.reshape(-1)[:h * w * 3].reshape(h, w, 3)
        results.append((rect, flat.copy()))

    import ctypes as C
    with pytest.raises(capi_err()):
        eng.pipeline_wait()                                   # nothing submitted
    for t in range(n_frames):
        if t == 5:
            eng.update(rig_b["scale"], rig_b["cameras"])      # takes effect for frames submitted from now on
        if t >= 3:
            wait_one()                                        # ring of three: free the oldest slot first
        (_, _, cw, ch), _ = eng.acquire()
        pano = panos[t % 3].array.reshape(-1)[:ch * cw * 3].reshape(ch, cw, 3)
        eng.pipeline_submit([p.array for p in pinned_frames[t]], pano)
    while len(results) < n_frames:
        wait_one()
    for t, (rect, pano) in enumerate(results):
        rig = rig_a if t < 5 else rig_b
        orect, opano, _ = H.oracle_compose(rig, host[t])
        assert rect == orect, t
        assert np.array_equal(pano, opano), "frame %d" % t
    eng.close()


def capi_err():
    from stitchingvideo_amd import _capi
    return _capi.StitchError


def test_profile_entry_points_report_every_launch():
    """mst_compose_profile(_batch): one HIP event between consecutive launches, names and durations per stage."""
    import torch
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    eng = H.make_engine(rig, max_batch=2)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(2)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    ptrs = [[t.data_ptr() for t in fr] for fr in dfr]
    strides = [[t.stride(0) for t in fr] for fr in dfr]
    torch.cuda.synchronize()
    single = eng.profile_stages(ptrs[0], strides[0], panos[0].data_ptr(), panos[0].stride(0))
    batch = eng.profile_stages_batch(ptrs, strides, [p.data_ptr() for p in panos], panos[0].stride(0))
    for stages in (single, batch):
        # one entry per launch, named as rocprofv3 names the kernel: the warp kernel(s), the pyramid levels above level 1
        # (streaming pyrDown and / or the LDS chain), the two canvas kernels
        names = [n for n, _ in stages]
        assert names[0].startswith("k_warp_pyr") and names[-2].startswith("k_blend_coarse<") and names[-1].startswith("k_blend_fine<")
        pyr = [n for n in names[1:-2] if not n.startswith("k_warp_pyr")]
        assert pyr and all(n.startswith(("k_pyrdown_stream<", "k_pyr_chain<")) for n in pyr)
        assert all(0.0 < ms < 50.0 for _, ms in stages)
    eng.synchronize()
    for i in range(2):
        assert np.array_equal(panos[i].cpu().numpy(), H.oracle_compose(rig, host[i])[1])
    call = eng.make_device_batch_call(ptrs, strides, [p.data_ptr() for p in panos], panos[0].stride(0))
    for p in panos:
        p.zero_()
    call(); call()
    eng.synchronize()
    assert np.array_equal(panos[1].cpu().numpy(), H.oracle_compose(rig, host[1])[1])
    eng.close()


@pytest.mark.parametrize("base_off,row_pad,mask_off,mask_pad", [(0, 1, 0, 1), (1, 2, 3, 0), (2, 3, 1, 5), (3, 0, 0, 0), (0, 4, 2, 2)])
def test_unaligned_panorama_and_mask_buffers(base_off, row_pad, mask_off, mask_pad):
    """A tight cv::Mat of a canvas whose width is not a multiple of four has rows (and, after the first, row starts)
    off the 4-byte grid: the fast level-0 path then writes its 12-byte runs unaligned.  Same panorama, bit for bit,
    and nothing outside the rows is touched."""
    import torch
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 4))            # several whole 128 x 32 tiles
    eng = H.make_engine(rig, max_batch=4)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    frames = rigmod.frames_for(rig, 3)
    dfr = [torch.from_numpy(f).to(dev) for f in frames]
    stride, mstride = cw * 3 + row_pad, cw + mask_pad
    pbuf = torch.full((base_off + ch * stride + 16,), 0x5A, dtype=torch.uint8, device=dev)
    mbuf = torch.full((mask_off + ch * mstride + 16,), 0x5A, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.compose_device([t.data_ptr() for t in dfr], [t.stride(0) for t in dfr], pbuf.data_ptr() + base_off, stride,
                       mbuf.data_ptr() + mask_off, mstride)
    eng.synchronize()
    _, opano, omask = H.oracle_compose(rig, frames)
    got = pbuf.cpu().numpy()
    rows = got[base_off:base_off + ch * stride].reshape(ch, stride)
    assert np.array_equal(rows[:, :cw * 3].reshape(ch, cw, 3), opano)
    assert (rows[:, cw * 3:] == 0x5A).all() and (got[:base_off] == 0x5A).all() and (got[base_off + ch * stride:] == 0x5A).all()
    gm = mbuf.cpu().numpy()
    mrows = gm[mask_off:mask_off + ch * mstride].reshape(ch, mstride)
    assert np.array_equal(mrows[:, :cw], omask)
    assert (mrows[:, cw:] == 0x5A).all() and (gm[:mask_off] == 0x5A).all() and (gm[mask_off + ch * mstride:] == 0x5A).all()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("blend", ["multiband", "feather", "no"])
def test_graph_replay_of_the_launch_sequence(blend):
    """MST_FLAG_GRAPH_REPLAY: the launch sequence of a call is captured once per set of buffers and replayed; a ring of
    two frame sets and two panoramas over several turns, a batch call, then a recalibration (new generation: new graphs)."""
    import torch
    cfg = rigmod.scaled_config("C3", 8)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    eng = H.make_engine(rig, max_batch=2, graph_replay=True)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(2)]
    exp = [H.oracle_compose(rig, fr) for fr in host]
    stream = torch.cuda.Stream(device=dev)
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    masks = [torch.zeros((ch, cw), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for turn in range(4):                                  # turn 0 captures two graphs, the others replay them
        for k in range(2):
            panos[k].zero_(); masks[k].zero_()
            torch.cuda.synchronize()
            eng.compose_device([t.data_ptr() for t in dfr[k]], [t.stride(0) for t in dfr[k]], panos[k].data_ptr(), panos[k].stride(0),
                               masks[k].data_ptr(), masks[k].stride(0), stream=stream.cuda_stream)
            stream.synchronize()
            assert np.array_equal(panos[k].cpu().numpy(), exp[k][1]), "turn %d slot %d" % (turn, k)
            assert np.array_equal(masks[k].cpu().numpy(), exp[k][2])
    for rep in range(2):                                   # a batch of both frame sets: its own graph
        for p in panos:
            p.zero_()
        torch.cuda.synchronize()
        eng.compose_device_batch([[t.data_ptr() for t in fr] for fr in dfr], [[t.stride(0) for t in fr] for fr in dfr],
                                 [p.data_ptr() for p in panos], panos[0].stride(0), stream=stream.cuda_stream)
        stream.synchronize()
        for k in range(2):
            assert np.array_equal(panos[k].cpu().numpy(), exp[k][1])
    # recalibration: the same buffers under a new calibration must not replay the old sequence
    rig2 = rigmod.build_rig(cfg)
    for c in rig2["cameras"]:
        if c.get("gain_map") is not None:
            c["gain_map"] = (np.asarray(c["gain_map"], np.float32) * np.float32(1.07)).astype(np.float32)
    eng.update(rig2["scale"], rig2["cameras"])
    exp2 = H.oracle_compose(rig2, host[0])
    for rep in range(2):
        panos[0].zero_()
        torch.cuda.synchronize()
        eng.compose_device([t.data_ptr() for t in dfr[0]], [t.stride(0) for t in dfr[0]], panos[0].data_ptr(), panos[0].stride(0),
                           stream=stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(panos[0].cpu().numpy(), exp2[1])
    eng.close()


@pytest.mark.parametrize("blend,lut", [("feather", False), ("no", True), ("multiband", False)])
def test_deep_overlap_batches_match_oracle(blend, lut):
    """mst_config.overlap_depth = 4 with batched calls, for the multi-band blend and for the single-pass composites: twelve calls of
    two frame sets each, panorama buffers in a rotation of eight; a recalibration lands in the middle (calls before it keep their
    state, calls after it use the new one)."""
    import torch
    cfg = rigmod.scaled_config("C2" if blend == "feather" else "C3", 4 if blend == "feather" else 8)
    cfg["blend"] = blend
    rig = rigmod.build_rig(cfg)
    cfg_b = dict(cfg, yaws=[y + 3.0 for y in cfg["yaws"]])
    rig_b = rigmod.build_rig(cfg_b)
    n = len(rig["cameras"])
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend=blend, expos=rig["expos"], driver_lut=lut, max_batch=2, overlap_frames=True, overlap_depth=4)
    eng.update(rig["scale"], rig["cameras"])
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(6)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]

    def want(r, i):
        if lut:
            ref = S.PanoramaEngine(n, warp=r["warp"], blend=blend, expos=r["expos"], driver_lut=True)
            ref.update(r["scale"], r["cameras"])
            out = ref.compose(host[i])[0]
            ref.close()
            return out
        return H.oracle_compose(r, host[i], blend=blend)[1]
    results = []
    for phase, r in enumerate((rig, rig_b)):
        if phase == 1:
            eng.update(r["scale"], r["cameras"])
        (_, _, cw, ch), _ = eng.acquire()
        panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(8)]
        torch.cuda.synchronize()
        order = []
        for call in range(6):
            a, b = (2 * call) % 6, (2 * call + 1) % 6
            pa, pb = panos[(2 * call) % 8], panos[(2 * call + 1) % 8]
            eng.compose_device_batch([[t.data_ptr() for t in dfr[a]], [t.data_ptr() for t in dfr[b]]],
                                     [[t.stride(0) for t in dfr[a]], [t.stride(0) for t in dfr[b]]], [pa.data_ptr(), pb.data_ptr()], pa.stride(0))
            order += [(a, pa), (b, pb)]
        eng.synchronize()
        last = {}
        for i, p in order:
            last[p.data_ptr()] = (i, p)                      # the call that wrote the buffer last
        results.append((r, [(i, p.cpu().numpy()) for i, p in last.values()]))
        eng.release()
    eng.close()
    for r, outs in results:
        cache = {}
        for i, got in outs:
            if i not in cache:
                cache[i] = want(r, i)
            assert np.array_equal(got, cache[i]), (blend, i, H.diff_report(got, cache[i]))


@pytest.mark.parametrize("blend,batch", [("multiband", 1), ("multiband", 2), ("feather", 1)])
def test_calls_in_flight_behind_a_producer_and_a_consumer_stream(blend, batch):
    """mst_compose_device_batch_ex with overlap_depth = 4: the frames of call t + 1 land by asynchronous H2D copies on a copy
    stream while call t runs, the panoramas leave by asynchronous D2H copies on a reader stream, and the panorama ring holds
    exactly overlap_depth buffers -- nothing but the two events per call orders a call (which runs on an engine-owned stream)
    behind its producer and its consumer.  200 calls, no host synchronisation in the loop, every panorama against the oracle.
    Both side streams are slowed down by a large device copy in front of every transfer, so that frames arrive late and
    panoramas are read late: a call that did not wait would read stale frames / overwrite a panorama still being read."""
    import torch
    depth, n_calls, distinct = 4, 200, 3
    rig = rigmod.build_rig(rigmod.scaled_config("C3" if blend == "multiband" else "C2", 8))
    eng = H.make_engine(rig, overlap_frames=True, overlap_depth=depth, max_batch=batch)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    n = len(rig["cameras"])
    host = [rigmod.frames_for(rig, i) for i in range(distinct)]
    expected = [H.oracle_compose(rig, fr)[1] for fr in host]
    pinned = [[torch.from_numpy(f).pin_memory() for f in fr] for fr in host]
    ring = depth                                   # frame and panorama slots: exactly as many as calls can be in flight
    d_frames = [[[torch.zeros_like(t, device=dev) for t in pinned[0]] for _ in range(batch)] for _ in range(ring)]
    d_panos = [[torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)] for _ in range(ring)]
    out = torch.zeros((n_calls, batch, ch, cw, 3), dtype=torch.uint8).pin_memory()
    ballast_a, ballast_b = (torch.empty(48 << 20, dtype=torch.uint8, device=dev) for _ in range(2))
    s_copy, s_call, s_read = (torch.cuda.Stream(device=dev) for _ in range(3))
    ev_ready = [torch.cuda.Event() for _ in range(ring)]
    ev_done = [torch.cuda.Event() for _ in range(ring)]
    ev_read = [torch.cuda.Event() for _ in range(ring)]
    torch.cuda.synchronize()
    for t in range(n_calls):
        k = t % ring
        with torch.cuda.stream(s_copy):
            if t >= ring:
                s_copy.wait_event(ev_done[k])       # the slot's frames were last read by call t - ring (the caller's duty, as ever)
            ballast_b.copy_(ballast_a, non_blocking=True)                      # (the frames arrive late)
            for f in range(batch):
                for i in range(n):
                    d_frames[k][f][i].copy_(pinned[(t + f) % distinct][i], non_blocking=True)
            ev_ready[k].record(s_copy)
        eng.compose_device_batch_ex([[x.data_ptr() for x in fs] for fs in d_frames[k]], [[x.stride(0) for x in fs] for fs in d_frames[k]],
                                    [p.data_ptr() for p in d_panos[k]], d_panos[k][0].stride(0), stream=s_call.cuda_stream,
                                    frames_ready=ev_ready[k].cuda_event, outputs_free=ev_read[k].cuda_event if t >= ring else None)
        ev_done[k].record(s_call)
        with torch.cuda.stream(s_read):
            s_read.wait_event(ev_done[k])
            ballast_b.copy_(ballast_a, non_blocking=True) if t % 3 == 0 else None  # (some panoramas are read late)
            for f in range(batch):
                out[t, f].copy_(d_panos[k][f], non_blocking=True)
            ev_read[k].record(s_read)
    torch.cuda.synchronize()
    got = out.numpy()
    for t in range(n_calls):
        for f in range(batch):
            assert np.array_equal(got[t, f], expected[(t + f) % distinct]), "call %d frame set %d: %s" % (t, f, H.diff_report(got[t, f], expected[(t + f) % distinct]))
    eng.close()


def test_calls_in_flight_order_writers_of_overlapping_views():
    """Two calls in flight whose panoramas are sub-views of ONE allocation that share rows (the second starts a few rows into the
    first) write overlapping bytes: the engine orders them like calls on the same pointer (byte intervals, not pointer equality),
    so the bytes both write hold the LATER call's pixels."""
    import torch
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    eng = H.make_engine(rig, overlap_frames=True, overlap_depth=4)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, i) for i in range(4)]
    exp = [H.oracle_compose(rig, fr)[1] for fr in host]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    shift = 5                                       # rows
    big = torch.zeros((ch + shift, cw, 3), dtype=torch.uint8, device=dev)
    views = [big[:ch], big[shift:shift + ch]]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for rep in range(10):
        for i in range(4):
            v = views[i % 2]
            eng.compose_device([t.data_ptr() for t in dfr[i]], [t.stride(0) for t in dfr[i]], v.data_ptr(), v.stride(0), stream=stream.cuda_stream)
    stream.synchronize()
    got = big.cpu().numpy()
    # the last call (i = 3) wrote view 1 whole; view 0's rows above the overlap still hold call i = 2
    assert np.array_equal(got[shift:], exp[3])
    assert np.array_equal(got[:shift], exp[2][:shift])
    eng.close()

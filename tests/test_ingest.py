"""Frame ingest over shared memory (include/mi355_ingest.h): the reference's DatePacket wire format and two-event
handshake (R64/stitching_sharemem.{h,cpp}) between a capture process and the stitcher."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from stitchingvideo_amd import ingest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SERVER = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from stitchingvideo_amd import ingest
name, shapes = sys.argv[1], eval(sys.argv[2])
with ingest.FrameServer(name, max(h * w * 3 for h, w in shapes)) as srv:
    print("ready", flush=True)
    for i, (h, w) in enumerate(shapes):
        frame = np.random.default_rng(100 + i).integers(0, 256, (h, w, 3)).astype(np.uint8)
        srv.publish(frame, path="rtsp://camera/%%d" %% i, timeout_ms=20000)
    sys.stdin.read()          # keep the segment alive until the client is done
""" % ROOT


def test_packet_layout_is_the_reference_struct():
    assert C.sizeof(ingest.FramePacket) == 120 + 4 * 4
    assert ingest.FramePacket.width.offset == 120 and ingest.FramePacket.buferPointer.offset == 132


def test_frames_cross_processes_bit_exact():
    name = "/mst_ingest_test_%d" % os.getpid()
    shapes = [(48, 64), (33, 47), (48, 64), (1, 1)]
    srv = subprocess.Popen([sys.executable, "-c", SERVER, name, repr(shapes)], stdin=subprocess.PIPE,
                           stdout=subprocess.PIPE, text=True)
    try:
        assert srv.stdout.readline().strip() == "ready"
        with ingest.FrameClient(name) as cli:
            buf = np.zeros((48, 64, 3), np.uint8)            # one Mat re-used for every frame, as DRV:739-741 does
            for i, (h, w) in enumerate(shapes):
                if (h, w) != buf.shape[:2]:
                    view, path = cli.get(np.zeros((h, w, 3), np.uint8), timeout_ms=20000)
                else:
                    view, path = cli.get(buf, timeout_ms=20000)
                want = np.random.default_rng(100 + i).integers(0, 256, (h, w, 3)).astype(np.uint8)
                assert view.shape == (h, w, 3) and np.array_equal(view, want)
                assert path == "rtsp://camera/%d" % i
            with pytest.raises(ingest.IngestError) as ei:    # nothing more is coming: the request is withdrawn
                cli.get(buf, timeout_ms=200)
            assert ei.value.code == ingest.ERR_TIMEOUT
            with pytest.raises(ingest.IngestError):          # a destination that cannot hold the frame
                ingest.FrameClient(name + "_missing")
    finally:
        srv.stdin.close()
        srv.wait(timeout=20)
    assert srv.returncode == 0
    with pytest.raises(ingest.IngestError):                  # the server unlinked its segment on close
        ingest.FrameClient(name)


def test_handshake_between_threads_and_small_destination():
    name = "/mst_ingest_thr_%d" % os.getpid()
    frames = [np.full((20, 30, 3), 7 * (i + 1), np.uint8) for i in range(5)]
    with ingest.FrameServer(name, 20 * 30 * 3) as srv:
        with pytest.raises(ingest.IngestError):              # larger than the segment was made for
            srv.publish(np.zeros((40, 30, 3), np.uint8), timeout_ms=0)
        with pytest.raises(ingest.IngestError) as ei:        # no client asked for a frame
            srv.publish(frames[0], timeout_ms=100)
        assert ei.value.code == ingest.ERR_TIMEOUT
        t = threading.Thread(target=lambda: [srv.publish(f, path="f%d" % i, timeout_ms=20000) for i, f in enumerate(frames)])
        t.start()
        with ingest.FrameClient(name) as cli:
            out = np.zeros((20, 30, 3), np.uint8)
            for i, f in enumerate(frames[:4]):
                view, path = cli.get(out, timeout_ms=20000)
                assert np.array_equal(view, f) and path == "f%d" % i
            with pytest.raises(ingest.IngestError):          # too small for the 20 x 30 frame (it is consumed anyway)
                cli.get(np.zeros((10, 30, 3), np.uint8), timeout_ms=20000)
        t.join()


@pytest.mark.gpu
def test_frame_lands_in_device_memory():
    import torch
    name = "/mst_ingest_gpu_%d" % os.getpid()
    frame = np.random.default_rng(5).integers(0, 256, (270, 480, 3)).astype(np.uint8)
    with ingest.FrameServer(name, frame.nbytes) as srv, ingest.FrameClient(name) as cli:
        t = threading.Thread(target=lambda: [srv.publish(frame, path="cam", timeout_ms=20000) for _ in range(2)])
        t.start()
        dev = torch.zeros((270, 480, 3), dtype=torch.uint8, device="cuda")
        assert cli.get_device(dev.data_ptr(), dev.stride(0), dev.numel())[:2] == (480, 270)
        assert np.array_equal(dev.cpu().numpy(), frame)
        padded = torch.zeros((270, 1500), dtype=torch.uint8, device="cuda")        # strided destination
        cli.get_device(padded.data_ptr(), padded.stride(0), padded.numel())
        assert np.array_equal(padded[:, :1440].cpu().numpy().reshape(270, 480, 3), frame)
        t.join()


def test_nv12_packets_cross_the_segment_and_say_their_format():
    name = "/mst_ingest_nv12_%d" % os.getpid()
    yuv = ingest.synth_nv12(0, 3, 64, 48)
    bgr = np.random.default_rng(1).integers(0, 256, (48, 64, 3)).astype(np.uint8)
    with ingest.FrameServer(name, 64 * 48 * 3) as srv:
        with pytest.raises(ingest.IngestError):              # odd sizes have no NV12 form
            srv.publish_nv12(np.zeros((3, 5), np.uint8), timeout_ms=0)
        t = threading.Thread(target=lambda: (srv.publish_nv12(yuv, path="nv", timeout_ms=20000), srv.publish(bgr, path="bgr", timeout_ms=20000)))
        t.start()
        with ingest.FrameClient(name) as cli:
            hdr = ingest.FramePacket()
            out = np.zeros((72, 64), np.uint8)
            L = ingest._lib()
            assert L.mst_shm_get(cli._h, C.byref(hdr), out.ctypes.data, 64, out.size, 20000) == 0
            assert (hdr.width, hdr.height, hdr.flag) == (64, 48, 64 * 48 * 3 // 2) and ingest.packet_format(hdr) == ingest.PIXEL_NV12
            assert np.array_equal(out, yuv)
            out3 = np.zeros((48, 64, 3), np.uint8)
            assert L.mst_shm_get(cli._h, C.byref(hdr), out3.ctypes.data, 64 * 3, out3.size, 20000) == 0
            assert ingest.packet_format(hdr) == ingest.PIXEL_BGR and np.array_equal(out3, bgr)
        t.join()
    bad = ingest.FramePacket(width=10, height=10, flag=123)
    with pytest.raises(ingest.IngestError):
        ingest.packet_format(bad)


def _spawn_producers(prefix, n, w, h, fmt, distinct, count):
    names = ["%s_%d" % (prefix, i) for i in range(n)]
    procs = [subprocess.Popen([sys.executable, "-m", "stitchingvideo_amd.ingest", names[i], str(i), str(w), str(h), fmt, str(distinct), str(count)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for i in range(n)]
    for p in procs:
        assert p.stdout.readline().strip() == "ready"
    return names, procs


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bgr", "nv12"])
def test_native_pipeline_from_capture_processes(fmt):
    """mst_pipeline_submit_shm: one native call per frame set -- capture processes -> shared memory -> the engine's host pipeline ->
    panorama in page-locked memory; every panorama equals the oracle's on the frames the producers sent (three in flight)."""
    import helpers as H
    import oracle_lib as O
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 4 if fmt == "nv12" else 8))
    n = len(rig["cameras"])
    w, h = rig["size"]
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend=rig["blend"], expos=rig["expos"], num_bands=5, frames_nv12=(fmt == "nv12"))
    eng.update(rig["scale"], rig["cameras"])
    n_sets = 8
    names, procs = _spawn_producers("/mst_nat_%s_%d" % (fmt, os.getpid()), n, w, h, fmt, 3, n_sets)
    try:
        st = ingest.PipelineIngest(eng, names)
        got = []
        for t in range(n_sets):
            if t >= st.depth:
                got.append(st.wait().copy())
            st.submit()
        while st.completed < st.submitted:
            got.append(st.wait().copy())
        st.close()
    finally:
        for p in procs:
            p.stdin.close()
        for p in procs:
            p.wait(timeout=30)
    assert len(got) == n_sets
    expected = {}
    for t, pano in enumerate(got):
        k = t % 3
        if k not in expected:
            frames = [O.cvt_nv12_bgr(ingest.synth_nv12(i, k, w, h)) if fmt == "nv12" else rigmod.synth_frame(i, k, w, h) for i in range(n)]
            expected[k] = H.oracle_compose(rig, frames)[1]
        assert np.array_equal(pano, expected[k]), (fmt, t)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prefetch,async_release", [(False, False), (True, False)])
def test_pull_modes_that_wait_for_the_bus(prefetch, async_release):
    """the two older forms of the pull (MST_SHM flags 0 and MST_SHM_PREFETCH: the call returns when the copies are complete) through the same loop"""
    import helpers as H
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    n = len(rig["cameras"])
    w, h = rig["size"]
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend=rig["blend"], expos=rig["expos"], num_bands=5, overlap_frames=True, overlap_depth=4)
    eng.update(rig["scale"], rig["cameras"])
    n_sets = 7
    names, procs = _spawn_producers("/mst_pm_%d_%d_%d" % (prefetch, async_release, os.getpid()), n, w, h, "bgr", 3, n_sets)
    try:
        st = ingest.IngestStitcher(eng, names, [c["size"] for c in rig["cameras"]], prefetch=prefetch, async_release=async_release)
        got = []
        for t in range(n_sets):
            if t >= 3:
                got.append(st.wait().copy())
            st.submit()
        while st.completed < st.submitted:
            got.append(st.wait().copy())
        st.close()
    finally:
        for p in procs:
            p.stdin.close()
        for p in procs:
            p.wait(timeout=30)
    expected = {}
    for t, pano in enumerate(got):
        k = t % 3
        if k not in expected:
            expected[k] = H.oracle_compose(rig, [rigmod.synth_frame(i, k, w, h) for i in range(n)])[1]
        assert np.array_equal(pano, expected[k]), (prefetch, t)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [0, 4])
@pytest.mark.parametrize("fmt", ["bgr", "nv12"])
def test_capture_processes_to_panorama_end_to_end(fmt, depth):
    """SURVEY 8 f3 as a whole: N capture processes -> shared memory -> HBM -> compose -> panorama in host memory, BGR and NV12
    packets; every panorama equals the oracle's on the frames the producers sent."""
    import helpers as H
    import oracle_lib as O
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 4 if fmt == "nv12" else 8))      # (NV12 needs even sizes: 480 x 270)
    n = len(rig["cameras"])
    w, h = rig["size"]
    # depth 4: the engine's calls-in-flight mode (every call whole on an engine-owned stream): IngestStitcher orders each call
    # behind its own frames and behind the copy-out of the panorama buffer it re-uses with events, not with host waits
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend=rig["blend"], expos=rig["expos"], num_bands=5, frames_nv12=(fmt == "nv12"),
                           overlap_frames=depth > 0, overlap_depth=depth)
    eng.update(rig["scale"], rig["cameras"])
    n_sets = 5 if depth == 0 else 9
    names, procs = _spawn_producers("/mst_e2e_%s_%d_%d" % (fmt, depth, os.getpid()), n, w, h, fmt, 3, n_sets)
    try:
        st = ingest.IngestStitcher(eng, names, [c["size"] for c in rig["cameras"]])
        got = []
        for t in range(n_sets):
            if t >= 3:
                got.append(st.wait().copy())
            st.submit()
        while st.completed < st.submitted:
            got.append(st.wait().copy())
        st.close()
    finally:
        for p in procs:
            p.stdin.close()
        for p in procs:
            p.wait(timeout=30)
    assert len(got) == n_sets
    expected = {}
    for t, pano in enumerate(got):
        k = t % 3
        if fmt == "nv12":
            frames = [O.cvt_nv12_bgr(ingest.synth_nv12(i, k, w, h)) for i in range(n)]
        else:
            frames = [rigmod.synth_frame(i, k, w, h) for i in range(n)]
        if k not in expected:
            expected[k] = H.oracle_compose(rig, frames)[1]
        assert np.array_equal(pano, expected[k]), (fmt, depth, t)
    eng.close()

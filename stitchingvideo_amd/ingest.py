"""ctypes mirror of include/mi355_ingest.h: the reference's shared-memory frame hand-off (Stitching_sharemem,
R64/stitching_sharemem.{h,cpp}) on POSIX shared memory.  FrameServer is the capture-process side (tests, demos),
FrameClient the stitcher side (ShareMemPre / GetMemData / SharememClose)."""
import ctypes as C

import numpy as np

from . import _capi as capi

ERR_TIMEOUT = -7
PIXEL_BGR, PIXEL_NV12 = 0, 1


class FramePacket(C.Structure):
    """struct DatePacket (stitching_sharemem.h:40-46)."""
    _fields_ = [("filePath", C.c_char * 120), ("width", C.c_int32), ("height", C.c_int32), ("flag", C.c_int32),
                ("buferPointer", C.c_int32)]


def _lib():
    L = capi.load()
    if not getattr(L, "_ingest_ready", False):
        vp = C.c_void_p
        L.mst_shm_serve.restype = C.c_int
        L.mst_shm_serve.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
        L.mst_shm_publish.restype = C.c_int
        L.mst_shm_publish.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp, C.c_size_t, C.c_int]
        L.mst_shm_attach.restype = C.c_int
        L.mst_shm_attach.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.mst_shm_get.restype = C.c_int
        L.mst_shm_get.argtypes = [vp, C.POINTER(FramePacket), vp, C.c_size_t, C.c_size_t, C.c_int]
        L.mst_shm_get_device.restype = C.c_int
        L.mst_shm_get_device.argtypes = [vp, C.POINTER(FramePacket), vp, C.c_size_t, C.c_size_t, vp, C.c_int]
        L.mst_shm_publish_nv12.restype = C.c_int
        L.mst_shm_publish_nv12.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp, C.c_size_t, C.c_int]
        L.mst_shm_packet_format.restype = C.c_int
        L.mst_shm_packet_format.argtypes = [C.POINTER(FramePacket)]
        L.mst_shm_get_device_multi.restype = C.c_int
        L.mst_shm_get_device_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(FramePacket), C.POINTER(vp), C.POINTER(C.c_size_t),
                                               C.POINTER(C.c_size_t), vp, C.c_int]
        L.mst_shm_get_device_multi_ex.restype = C.c_int
        L.mst_shm_get_device_multi_ex.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(FramePacket), C.POINTER(vp), C.POINTER(C.c_size_t),
                                                  C.POINTER(C.c_size_t), vp, C.c_int, C.c_int]
        L.mst_shm_close.restype = None
        L.mst_shm_close.argtypes = [vp]
        L._ingest_ready = True
    return L


class IngestError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed (%d)" % (what, code))
        self.code = code


class _Handle:
    def __init__(self):
        self._L = _lib()
        self._h = C.c_void_p()

    def close(self):
        if self._h:
            self._L.mst_shm_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class FrameServer(_Handle):
    def __init__(self, name, max_frame_bytes):
        super().__init__()
        rc = self._L.mst_shm_serve(name.encode(), max_frame_bytes, C.byref(self._h))
        if rc:
            raise IngestError(rc, "mst_shm_serve")

    def publish(self, frame, path="", timeout_ms=-1):
        frame = np.ascontiguousarray(frame, np.uint8)
        h, w = frame.shape[:2]
        rc = self._L.mst_shm_publish(self._h, path.encode(), w, h, frame.ctypes.data, frame.strides[0], timeout_ms)
        if rc:
            raise IngestError(rc, "mst_shm_publish")


    def publish_nv12(self, yuv, path="", timeout_ms=-1):
        """An NV12 frame: (h * 3 / 2, w) uint8, h rows of Y then h / 2 rows of interleaved U V."""
        yuv = np.ascontiguousarray(yuv, np.uint8)
        h, w = yuv.shape[0] * 2 // 3, yuv.shape[1]
        rc = self._L.mst_shm_publish_nv12(self._h, path.encode(), w, h, yuv.ctypes.data, yuv.strides[0], timeout_ms)
        if rc:
            raise IngestError(rc, "mst_shm_publish_nv12")


def packet_format(hdr):
    """PIXEL_BGR / PIXEL_NV12 of a received FramePacket (its `flag` byte count says which)."""
    rc = _lib().mst_shm_packet_format(C.byref(hdr))
    if rc < 0:
        raise IngestError(rc, "mst_shm_packet_format")
    return rc


SHM_PREFETCH, SHM_ASYNC = 1, 2


def get_device_multi(clients, device_ptrs, strides, capacities, stream=None, timeout_ms=-1, prefetch=False, async_release=False):
    """One frame from every client straight into device memory (mst_shm_get_device_multi_ex): all servers are asked first
    and fill their segments concurrently; a camera's copy is queued when its packet lands; prefetch: each camera is asked for
    its next frame the moment its copy has left the segment -> list of (width, height, format, path)."""
    n = len(clients)
    L = _lib()
    hs = (C.c_void_p * n)(*[c._h for c in clients])
    hdr = (FramePacket * n)()
    dp = (C.c_void_p * n)(*device_ptrs)
    st = (C.c_size_t * n)(*strides)
    cp = (C.c_size_t * n)(*capacities)
    rc = L.mst_shm_get_device_multi_ex(hs, n, hdr, dp, st, cp, stream, timeout_ms, (SHM_PREFETCH if prefetch else 0) | (SHM_ASYNC if async_release else 0))
    if rc:
        raise IngestError(rc, "mst_shm_get_device_multi_ex")
    return [(h.width, h.height, packet_format(h), h.filePath.decode(errors="replace")) for h in hdr]


class FrameClient(_Handle):
    def __init__(self, name):
        super().__init__()
        rc = self._L.mst_shm_attach(name.encode(), C.byref(self._h))
        if rc:
            raise IngestError(rc, "mst_shm_attach")

    def get(self, out, timeout_ms=-1):
        """GetMemData into `out` ((H, W, 3) uint8, at least as large as the frame; the reference re-uses the
        cv::Mat it is handed the same way) -> (view of the frame inside `out`, path)."""
        hdr = FramePacket()
        if out.dtype != np.uint8 or out.ndim != 3 or out.shape[2] != 3 or out.strides[2] != 1 or out.strides[1] != 3:
            raise ValueError("out must be a (H, W, 3) uint8 array with packed pixels")
        rc = self._L.mst_shm_get(self._h, C.byref(hdr), out.ctypes.data, out.strides[0], out.strides[0] * out.shape[0],
                                 timeout_ms)
        if rc:
            raise IngestError(rc, "mst_shm_get")
        return out[:hdr.height, :hdr.width], hdr.filePath.decode(errors="replace")

    def get_device(self, device_ptr, stride, capacity, stream=None, timeout_ms=-1):
        """GetMemData straight into device memory -> (width, height, path)."""
        hdr = FramePacket()
        rc = self._L.mst_shm_get_device(self._h, C.byref(hdr), device_ptr, stride, capacity, stream, timeout_ms)
        if rc:
            raise IngestError(rc, "mst_shm_get_device")
        return hdr.width, hdr.height, hdr.filePath.decode(errors="replace")


# ---- the whole ingest path: capture processes -> shared memory -> HBM -> compose -> panorama in host memory ---------
def synth_nv12(cam, frame_index, w, h):
    """A deterministic synthetic NV12 frame ((h * 3 / 2, w) uint8) for tests and the bench's producers."""
    rng = np.random.default_rng(1000003 * (cam + 1) + frame_index)
    return rng.integers(16, 236, size=(h * 3 // 2, w)).astype(np.uint8)


def producer_main(argv):
    """A capture process (GetFromRtsp.exe's role): `python -m stitchingvideo_amd.ingest NAME CAM W H FORMAT DISTINCT COUNT`
    serves COUNT frames (0: until killed) of DISTINCT pre-generated synthetic frames of camera CAM over segment NAME.
    Prints "ready" once the segment exists."""
    import sys
    from . import rig as rigmod
    name, cam, w, h, fmt, distinct, count = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4], int(argv[5]), int(argv[6])
    nv12 = fmt == "nv12"
    frames = [synth_nv12(cam, k, w, h) if nv12 else rigmod.synth_frame(cam, k, w, h) for k in range(distinct)]
    with FrameServer(name, w * h * 3) as srv:
        print("ready", flush=True)
        k = 0
        while count == 0 or k < count:
            try:
                (srv.publish_nv12 if nv12 else srv.publish)(frames[k % distinct], path="synthetic://cam%d/%d" % (cam, k), timeout_ms=5000)
            except IngestError as e:
                if e.code != ERR_TIMEOUT:
                    raise
                if count == 0:
                    continue        # (idle: keep serving until the parent closes our stdin)
                break
            k += 1
        sys.stdin.read()            # keep the segment alive until the consumer is done


class IngestStitcher:
    """The reference's frame loop (DRV:736-763) fed the reference's way -- every camera's frame pulled from its capture
    process over shared memory (Stitching_sharemem::GetMemData, DRV:739-740) -- as a three-stage pipeline on the device:
    stage 1 pulls frame set t + 1 of all cameras into HBM (the capture processes fill their segments concurrently, the
    copies are DMA out of the page-locked segments), stage 2 composes frame set t, stage 3 copies panorama t - 1 into
    page-locked host memory.  BGR or NV12 packets (the engine's MST_FLAG_FRAMES_NV12 says which it composes from)."""

    def __init__(self, engine, names, sizes, device=0, depth=3, prefetch=True, async_release=True):
        """engine: a calibrated PanoramaEngine; names: one segment name per camera; sizes: the cameras' (width, height);
        prefetch: ask every camera for its next frame as soon as its copy has left the segment (MST_SHM_PREFETCH)."""
        self.prefetch = bool(prefetch)
        # async_release (MST_SHM_ASYNC): submit() does not wait for the bus -- the copies, the compose and the copy-out of a frame set
        # are all queued at once and ordered by events; the capture processes are released by host functions behind the copies
        self.async_release = bool(async_release)
        import torch
        self._torch = torch
        self.eng = engine
        self.clients = [FrameClient(n) for n in names]
        self.dev = torch.device("cuda", device)
        (self.cx, self.cy, self.cw, self.ch), _ = engine.acquire()
        self.nv12 = bool(getattr(engine, "_frames_nv12", False))
        self.depth = depth
        self.sizes = [(int(w), int(h)) for (w, h) in sizes]
        self.sets = [[torch.empty((h * 3 // 2, w) if self.nv12 else (h, w, 3), dtype=torch.uint8, device=self.dev) for (w, h) in self.sizes]
                     for _ in range(depth)]
        self.d_panos = [torch.empty((self.ch, self.cw, 3), dtype=torch.uint8, device=self.dev) for _ in range(depth)]
        self.h_panos = [torch.empty((self.ch, self.cw, 3), dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
        self.s_in, self.s_c, self.s_out = (torch.cuda.Stream(device=self.dev) for _ in range(3))
        self.ev_in = [torch.cuda.Event() for _ in range(depth)]
        self.ev_c = [torch.cuda.Event() for _ in range(depth)]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]
        self.submitted = self.completed = 0
        self.bytes_in = sum(t.numel() for t in self.sets[0])

    def submit(self, timeout_ms=20000):
        """Pull one frame set from the capture processes and queue its compose and the panorama's copy-out."""
        torch = self._torch
        k = self.submitted % self.depth
        if self.submitted - self.completed >= self.depth:
            raise RuntimeError("ring full: call wait() first")
        fs = self.sets[k]
        if self.async_release and self.submitted >= self.depth:
            self.s_in.wait_event(self.ev_c[k])          # the compose that last read this slot's frames (nobody waited for it on the host)
        # (without async_release stage 1 is complete on return: the servers may overwrite their segments at the next request; the slot's frames
        # were last read by the compose of `depth` submits ago, whose completion wait() has observed)
        got = get_device_multi(self.clients, [t.data_ptr() for t in fs], [t.stride(0) for t in fs], [t.numel() for t in fs],
                               stream=self.s_in.cuda_stream, timeout_ms=timeout_ms, prefetch=self.prefetch, async_release=self.async_release)
        want = PIXEL_NV12 if self.nv12 else PIXEL_BGR
        for (w, h, fmt, _), (ew, eh) in zip(got, self.sizes):
            if fmt != want or (w, h) != (ew, eh):
                raise IngestError(-2, "packet of %dx%d format %d where the calibration expects %dx%d format %d" % (w, h, fmt, ew, eh, want))
        self.ev_in[k].record(self.s_in)
        # The compose call waits for ITS frames (the copies just queued on the ingest stream) and for the copy-out that last
        # read this slot's panorama buffer -- explicit events (mst_compose_device_batch_ex), so that an engine with calls in
        # flight (overlap_depth >= 3: the call runs on an engine-owned stream, not ordered behind the compose stream) needs no
        # host synchronisation per frame set; an engine without them waits for the same events on the compose stream.
        out_free = self.ev_out[k].cuda_event if self.submitted >= self.depth else None
        self.eng.compose_device_batch_ex([[t.data_ptr() for t in fs]], [[t.stride(0) for t in fs]], [self.d_panos[k].data_ptr()],
                                         self.d_panos[k].stride(0), stream=self.s_c.cuda_stream,
                                         frames_ready=self.ev_in[k].cuda_event, outputs_free=out_free)
        self.ev_c[k].record(self.s_c)
        self.s_out.wait_event(self.ev_c[k])
        with torch.cuda.stream(self.s_out):
            self.h_panos[k].copy_(self.d_panos[k], non_blocking=True)
        self.ev_out[k].record(self.s_out)
        self.submitted += 1

    def wait(self):
        """The oldest submitted panorama, complete in page-locked host memory (valid until `depth` more submits)."""
        if self.completed >= self.submitted:
            raise RuntimeError("wait() without a submitted frame set")
        k = self.completed % self.depth
        self.ev_out[k].synchronize()
        self.completed += 1
        return self.h_panos[k].numpy()

    def close(self):
        self._torch.cuda.synchronize(self.dev)
        for c in self.clients:
            c.close()


class PipelineIngest:
    """The same frame loop through ONE native call per frame set (mst_pipeline_submit_shm): the engine's own host pipeline -- its
    copy-in, compute and copy-out streams, a ring of MST_PIPELINE_DEPTH device frame sets -- fed from the capture processes' segments.
    No torch: what a C++ host links."""
    depth = 3

    def __init__(self, engine, names):
        from .engine import PinnedArray
        self.eng = engine
        self.clients = [FrameClient(n) for n in names]
        (_, _, self.cw, self.ch), _ = engine.acquire()
        self._panos = [PinnedArray((self.ch, self.cw, 3)) for _ in range(self.depth)]
        self.submitted = self.completed = 0

    def submit(self, timeout_ms=20000):
        if self.submitted - self.completed >= self.depth:
            raise RuntimeError("ring full: call wait() first")
        self.eng.pipeline_submit_shm(self.clients, self._panos[self.submitted % self.depth].array, timeout_ms)
        self.submitted += 1

    def wait(self):
        """The oldest submitted panorama, complete in page-locked host memory (valid until `depth` more submits)."""
        self.eng.pipeline_wait()
        k = self.completed % self.depth
        self.completed += 1
        return self._panos[k].array

    def close(self):
        while self.completed < self.submitted:
            self.wait()
        self.eng.synchronize() if hasattr(self.eng, "synchronize") else None
        for c in self.clients:
            c.close()


if __name__ == "__main__":
    import sys as _sys
    producer_main(_sys.argv[1:])

"""ctypes mirror of include/mi355_ingest.h: the reference's shared-memory frame hand-off (Stitching_sharemem,
R64/stitching_sharemem.{h,cpp}) on POSIX shared memory.  FrameServer is the capture-process side (tests, demos),
FrameClient the stitcher side (ShareMemPre / GetMemData / SharememClose)."""
import ctypes as C

import numpy as np

from . import _capi as capi

ERR_TIMEOUT = -7


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

"""ctypes binding of libmi355stitch.so -- the C ABI declared in include/mi355_stitch.h.

The library is the product: hand-written HIP kernels for gfx950 behind plain-C entry points.
There is no Python or CPU fallback; if the shared object is missing the import fails loudly.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MST_LIB") or os.path.join(_HERE, "libmi355stitch.so")   # MST_LIB: tuning variants

MST_OK = 0
ERR_INVALID_ARG, ERR_NOT_CALIBRATED, ERR_HIP, ERR_NO_DEVICE, ERR_NEED_MORE_IMGS, ERR_OUT_OF_MEMORY = \
    -1, -2, -3, -4, -5, -6
WARP_PLANE, WARP_SPHERICAL, WARP_CYLINDRICAL = 0, 1, 2
(WARP_FISHEYE, WARP_STEREOGRAPHIC, WARP_COMPRESSED_PLANE, WARP_COMPRESSED_PLANE_PORTRAIT, WARP_PANINI,
 WARP_PANINI_PORTRAIT, WARP_MERCATOR, WARP_TRANSVERSE_MERCATOR, WARP_SPHERICAL_PORTRAIT,
 WARP_CYLINDRICAL_PORTRAIT, WARP_PLANE_PORTRAIT) = range(3, 14)
WARP_A2B1, WARP_A1_5B1, WARP_AB_CUSTOM = 1 << 8, 2 << 8, 3 << 8
# the reference's --warp names (CLI:102) plus the three detail:: portrait warpers that have none
WARP_NAMES = {
    "plane": 0, "spherical": 1, "cylindrical": 2, "fisheye": 3, "stereographic": 4,
    "compressedPlaneA2B1": 5 | WARP_A2B1, "compressedPlaneA1.5B1": 5 | WARP_A1_5B1,
    "compressedPlanePortraitA2B1": 6 | WARP_A2B1, "compressedPlanePortraitA1.5B1": 6 | WARP_A1_5B1,
    "paniniA2B1": 7 | WARP_A2B1, "paniniA1.5B1": 7 | WARP_A1_5B1,
    "paniniPortraitA2B1": 8 | WARP_A2B1, "paniniPortraitA1.5B1": 8 | WARP_A1_5B1,
    "mercator": 9, "transverseMercator": 10,
    "sphericalPortrait": 11, "cylindricalPortrait": 12, "planePortrait": 13,
}
BLEND_NO, BLEND_FEATHER, BLEND_MULTI_BAND = 0, 1, 2
EXPOS_NO, EXPOS_GAIN, EXPOS_GAIN_BLOCKS = 0, 1, 2
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
FLAG_DRIVER_LUT = 1
FLAG_DEBUG = 2
FLAG_OVERLAP_FRAMES = 4
FLAG_LENS_COMPOSED = 8
FLAG_FRAMES_NV12 = 16
FLAG_GRAPH_REPLAY = 32
MAX_CAMERAS, MAX_BANDS = 16, 10
DBG_CAM_WARPED_S16, DBG_CAM_GAUSS_S16, DBG_CAM_WEIGHT_F32, DBG_BAND_WEIGHT_F32, DBG_CANVAS_S16, DBG_CAM_NEED_U8 = \
    1, 2, 3, 4, 5, 6

EXPORTS = [
    "mst_warp_kind_from_name", "mst_create", "mst_destroy", "mst_last_error", "mst_update", "mst_acquire_calibration", "mst_release_calibration", "mst_canvas", "mst_strip", "mst_camera_roi", "mst_camera_mask",
    "mst_num_bands", "mst_compose", "mst_compose_device", "mst_compose_device_batch", "mst_compose_device_batch_ex", "mst_build_info", "mst_compose_profile",
    "mst_compose_profile_batch", "mst_synchronize", "mst_warp_roi",
    "mst_host_alloc", "mst_host_free", "mst_pipeline_submit", "mst_pipeline_wait", "mst_pipeline_pending",
    "mst_stage_cameras", "mst_stage_canvas", "mst_pyramid_buffer",
    "mst_payload_bytes", "mst_payload_pack", "mst_payload_unpack", "mst_stage_select_set",
    "mst_build_maps", "mst_build_maps_host", "mst_warp", "mst_exposure_apply",
    "mst_init_undistort_rectify_map", "mst_compose_lens_maps", "mst_cvt_nv12_bgr", "mst_remap_fixed",
    "mst_calibration_blob_size", "mst_calibration_save", "mst_calibration_load",
    "mst_multiband_geometry", "mst_debug_read", "mst_synth_frame", "mst_gain_map_size", "mst_exposure_feed", "mst_seam_voronoi", "mst_seam_dp",
    # include/mi355_ingest.h
    "mst_shm_serve", "mst_shm_publish", "mst_shm_publish_nv12", "mst_shm_packet_format", "mst_shm_attach", "mst_shm_get",
    "mst_shm_get_device", "mst_shm_get_device_multi", "mst_shm_get_device_multi_ex", "mst_pipeline_submit_shm", "mst_shm_close",
]


class Build(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("version", C.c_int32), ("timing_build", C.c_uint32),
                ("level0_packed", C.c_int32), ("description", C.c_char * 160)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("num_cameras", C.c_int32), ("warp_kind", C.c_int32),
                ("blend_kind", C.c_int32), ("num_bands", C.c_int32), ("feather_sharpness", C.c_float),
                ("expos_kind", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("upblack", C.c_float), ("downblack", C.c_float), ("leftblack", C.c_float),
                ("rightblack", C.c_float), ("max_batch", C.c_int32), ("weight_type", C.c_int32),
                ("warp_a", C.c_float), ("warp_b", C.c_float), ("strip_index", C.c_int32), ("strip_count", C.c_int32),
                ("overlap_depth", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("src_width", C.c_int32), ("src_height", C.c_int32), ("K", C.c_float * 9),
                ("R", C.c_float * 9), ("T", C.c_float * 3), ("gain_map", C.c_void_p),
                ("gain_w", C.c_int32), ("gain_h", C.c_int32), ("gain", C.c_float),
                ("mask", C.c_void_p), ("mask_width", C.c_int32), ("mask_height", C.c_int32),
                ("mask_stride", C.c_size_t), ("undistort_map1", C.c_void_p), ("undistort_map2", C.c_void_p),
                ("seam_mask", C.c_void_p), ("seam_mask_width", C.c_int32), ("seam_mask_height", C.c_int32),
                ("seam_mask_stride", C.c_size_t)]


class Calibration(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("num_cameras", C.c_int32),
                ("warped_image_scale", C.c_float), ("cameras", C.POINTER(Camera))]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]

    def tuple(self):
        return (self.x, self.y, self.width, self.height)


class StitchError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mi355stitch error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """dlopen the product library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C stitchingvideo_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # PyTorch's ROCm wheel ships its own HIP runtime under the soname this library links; the copy that is loaded
    # first serves the whole process, and torch only finds its GPUs with its own.  A process that will also use
    # torch (tests, bench.py, the sharded launcher) therefore needs torch loaded before this library.
    import importlib.util
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:  # a broken torch install must not take the engine down with it
            pass
    L = C.CDLL(LIB_PATH)
    fp, vp, ip = C.POINTER(C.c_float), C.c_void_p, C.POINTER(C.c_int)
    L.mst_warp_kind_from_name.restype = C.c_int
    L.mst_warp_kind_from_name.argtypes = [C.c_char_p]
    L.mst_create.restype = C.c_int
    L.mst_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.mst_destroy.restype = None
    L.mst_destroy.argtypes = [vp]
    L.mst_last_error.restype = C.c_char_p
    L.mst_last_error.argtypes = [vp]
    L.mst_update.restype = C.c_int
    L.mst_update.argtypes = [vp, C.POINTER(Calibration)]
    L.mst_acquire_calibration.restype = C.c_int
    L.mst_acquire_calibration.argtypes = [vp, C.POINTER(Rect), C.POINTER(C.c_uint64)]
    L.mst_strip.restype = C.c_int
    L.mst_strip.argtypes = [vp, C.POINTER(Rect)]
    L.mst_release_calibration.restype = C.c_int
    L.mst_release_calibration.argtypes = [vp]
    L.mst_canvas.restype = C.c_int
    L.mst_canvas.argtypes = [vp, C.POINTER(Rect)]
    L.mst_camera_roi.restype = C.c_int
    L.mst_camera_roi.argtypes = [vp, C.c_int, C.POINTER(Rect)]
    L.mst_camera_mask.restype = C.c_int
    L.mst_camera_mask.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.mst_num_bands.restype = C.c_int
    L.mst_num_bands.argtypes = [vp, ip]
    L.mst_compose.restype = C.c_int
    L.mst_compose.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), vp, C.c_size_t, vp, C.c_size_t]
    L.mst_compose_device.restype = C.c_int
    L.mst_compose_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), vp, C.c_size_t, vp,
                                     C.c_size_t, vp]
    L.mst_compose_device_batch.restype = C.c_int
    L.mst_compose_device_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                                           C.c_size_t, C.POINTER(vp), C.c_size_t, vp]
    L.mst_compose_device_batch_ex.restype = C.c_int
    L.mst_compose_device_batch_ex.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                                              C.c_size_t, C.POINTER(vp), C.c_size_t, vp, vp, vp]
    L.mst_build_info.restype = C.c_int
    L.mst_build_info.argtypes = [C.POINTER(Build)]
    L.mst_compose_profile_batch.restype = C.c_int
    L.mst_compose_profile_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                                            C.c_size_t, vp, C.c_int, vp, fp, ip]
    L.mst_calibration_blob_size.restype = C.c_size_t
    L.mst_calibration_blob_size.argtypes = [C.POINTER(Calibration)]
    L.mst_calibration_save.restype = C.c_int
    L.mst_calibration_save.argtypes = [C.POINTER(Calibration), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mst_calibration_load.restype = C.c_int
    L.mst_calibration_load.argtypes = [vp, C.c_size_t, C.POINTER(Calibration), C.POINTER(Camera), C.c_int]
    L.mst_cvt_nv12_bgr.restype = C.c_int
    L.mst_cvt_nv12_bgr.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]
    L.mst_compose_lens_maps.restype = C.c_int
    L.mst_compose_lens_maps.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_int, C.c_int]
    L.mst_init_undistort_rectify_map.restype = C.c_int
    L.mst_init_undistort_rectify_map.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    L.mst_remap_fixed.restype = C.c_int
    L.mst_remap_fixed.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_size_t, vp, vp, vp, C.c_size_t]
    L.mst_host_alloc.restype = vp
    L.mst_host_alloc.argtypes = [C.c_size_t]
    L.mst_host_free.restype = None
    L.mst_host_free.argtypes = [vp]
    L.mst_pipeline_submit.restype = C.c_int
    L.mst_pipeline_submit.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), vp, C.c_size_t]
    L.mst_pipeline_wait.restype = C.c_int
    L.mst_pipeline_wait.argtypes = [vp, C.POINTER(vp), C.POINTER(Rect)]
    L.mst_pipeline_pending.restype = C.c_int
    L.mst_pipeline_pending.argtypes = [vp]
    L.mst_synchronize.restype = C.c_int
    L.mst_synchronize.argtypes = [vp]
    L.mst_stage_cameras.restype = C.c_int
    L.mst_stage_cameras.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t), vp]
    L.mst_stage_canvas.restype = C.c_int
    L.mst_stage_canvas.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.mst_payload_bytes.restype = C.c_int
    L.mst_payload_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.mst_payload_pack.restype = C.c_int
    L.mst_payload_pack.argtypes = [vp, C.c_int, vp, vp]
    L.mst_payload_unpack.restype = C.c_int
    L.mst_payload_unpack.argtypes = [vp, C.c_int, vp, vp]
    L.mst_stage_select_set.restype = C.c_int
    L.mst_stage_select_set.argtypes = [vp, C.c_int]
    L.mst_pyramid_buffer.restype = C.c_int
    L.mst_pyramid_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.mst_compose_profile.restype = C.c_int
    L.mst_compose_profile.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), vp, C.c_size_t, vp, C.c_int,
                                      vp, fp, ip]
    L.mst_warp_roi.restype = C.c_int
    L.mst_warp_roi.argtypes = [C.c_int, C.c_float, fp, fp, fp, C.c_int, C.c_int, C.POINTER(Rect)]
    L.mst_build_maps.restype = C.c_int
    L.mst_build_maps.argtypes = [C.c_int, C.c_int, C.c_float, fp, fp, fp, C.c_int, C.c_int, vp, vp]
    L.mst_build_maps_host.restype = C.c_int
    L.mst_build_maps_host.argtypes = [C.c_int, C.c_float, fp, fp, fp, C.c_int, C.c_int, vp, vp]
    L.mst_warp.restype = C.c_int
    L.mst_warp.argtypes = [C.c_int, C.c_int, C.c_float, fp, fp, fp, vp, C.c_int, C.c_int, C.c_size_t,
                           C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    L.mst_exposure_apply.restype = C.c_int
    L.mst_exposure_apply.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int,
                                     C.c_int, C.c_float]
    L.mst_seam_dp.restype = C.c_int
    L.mst_seam_dp.argtypes = [C.c_int, C.c_int, ip, ip, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int, C.POINTER(vp),
                              C.POINTER(C.c_size_t), C.c_int]
    L.mst_seam_voronoi.restype = C.c_int
    L.mst_seam_voronoi.argtypes = [C.c_int, C.c_int, ip, ip, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.mst_gain_map_size.restype = C.c_int
    L.mst_gain_map_size.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip, ip]
    L.mst_exposure_feed.restype = C.c_int
    L.mst_exposure_feed.argtypes = [C.c_int, C.c_int, C.c_int, ip, C.POINTER(vp), C.POINTER(C.c_size_t), ip, C.POINTER(vp),
                                    C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(vp)]
    L.mst_multiband_geometry.restype = C.c_int
    L.mst_multiband_geometry.argtypes = [Rect, C.c_int, Rect, C.POINTER(Rect), ip, ip]
    L.mst_debug_read.restype = C.c_int
    L.mst_debug_read.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, ip, ip]
    L.mst_synth_frame.restype = C.c_int
    L.mst_synth_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    _lib = L
    return L


def build_info():
    """mst_build_info as a dict: {'version', 'timing_build', 'level0_packed', 'description'}."""
    b = Build()
    b.struct_size = C.sizeof(Build)
    check(load().mst_build_info(C.byref(b)))
    return {"version": b.version, "timing_build": int(b.timing_build), "level0_packed": b.level0_packed,
            "description": b.description.decode()}


def check(rc, handle=None):
    if rc != MST_OK:
        msg = load().mst_last_error(handle)
        raise StitchError(rc, msg.decode("utf-8", "replace") if msg else "")

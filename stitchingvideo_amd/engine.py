"""Host-side mirror of the reference's per-frame interfaces, bound to the C ABI.

Names, argument meaning and error behaviour follow the reference (R64 =
/root/reference/stitching/OpenCV2.4.11-Stitching-64bit/OpenCV2.4.11-Stitching):

  RotationWarper.warpRoi / buildMaps / warp     detail/warpers.hpp:53-72, warpers_inl.hpp:62-140
  ExposureCompensator.apply                     detail/exposure_compensate.hpp:51-64
  PanoramaEngine.update / compose               UpdateMat + StitchingAll, R64/stitching_detailed.cpp:696-770;
                                                Stitcher::composePanorama, R64/src/stitcher.cpp:264-313

Every pixel is computed by the HIP kernels in libmi355stitch.so; numpy is only the container
for host images (a cv::Mat stand-in: data + step).
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def _f9(a):
    return (C.c_float * 9)(*np.asarray(a, np.float32).reshape(-1))


def _f3(a):
    if a is None:
        a = (0.0, 0.0, 0.0)
    return (C.c_float * 3)(*np.asarray(a, np.float32).reshape(-1))


class RotationWarper:
    """cv::detail::RotationWarper: every projector of detail/warpers.hpp (plane / spherical / cylindrical maps are
    evaluated on the device, the others are built on the host per calibration and read by the kernels)."""

    kind = None

    def __init__(self, scale, device=0, kind=None):
        self.scale = float(scale)
        self.device = device
        if kind is not None:
            self.kind = capi.WARP_NAMES[kind] if isinstance(kind, str) else int(kind)

    def warpRoi(self, src_size, K, R, T=None):
        """src_size = (width, height) -> (x, y, width, height).  Host only."""
        r = capi.Rect()
        capi.check(capi.load().mst_warp_roi(self.kind, self.scale, _f9(K), _f9(R), _f3(T),
                                            int(src_size[0]), int(src_size[1]), C.byref(r)))
        return r.tuple()

    def buildMaps(self, src_size, K, R, T=None, host=False):
        """-> (roi, xmap, ymap); xmap/ymap are CV_32F of the roi size."""
        roi = self.warpRoi(src_size, K, R, T)
        xmap = np.empty((roi[3], roi[2]), np.float32)
        ymap = np.empty((roi[3], roi[2]), np.float32)
        L = capi.load()
        if host:
            rc = L.mst_build_maps_host(self.kind, self.scale, _f9(K), _f9(R), _f3(T), int(src_size[0]),
                                       int(src_size[1]), xmap.ctypes.data, ymap.ctypes.data)
        else:
            rc = L.mst_build_maps(self.device, self.kind, self.scale, _f9(K), _f9(R), _f3(T),
                                  int(src_size[0]), int(src_size[1]), xmap.ctypes.data, ymap.ctypes.data)
        capi.check(rc)
        return roi, xmap, ymap

    def warp(self, src, K, R, interp_mode=capi.INTER_LINEAR, border_mode=capi.BORDER_REFLECT, T=None):
        """-> (corner (x, y), dst).  src is CV_8UC1 (H, W) or CV_8UC3 (H, W, 3)."""
        src = np.asarray(src)
        if src.dtype != np.uint8 or src.ndim not in (2, 3) or (src.ndim == 3 and src.shape[2] != 3):
            raise capi.StitchError(capi.ERR_INVALID_ARG, "warp expects CV_8UC1 or CV_8UC3")
        if src.strides[-1] != 1 or (src.ndim == 3 and src.strides[1] != 3):
            src = np.ascontiguousarray(src)
        h, w = src.shape[:2]
        cn = 1 if src.ndim == 2 else 3
        roi = self.warpRoi((w, h), K, R, T)
        dst = np.empty((roi[3], roi[2]) + (() if cn == 1 else (3,)), np.uint8)
        capi.check(capi.load().mst_warp(self.device, self.kind, self.scale, _f9(K), _f9(R), _f3(T),
                                        src.ctypes.data, w, h, src.strides[0], cn, interp_mode, border_mode,
                                        dst.ctypes.data, dst.strides[0]))
        return (roi[0], roi[1]), dst


class PlaneWarper(RotationWarper):
    kind = capi.WARP_PLANE


class SphericalWarper(RotationWarper):
    kind = capi.WARP_SPHERICAL


class CylindricalWarper(RotationWarper):
    kind = capi.WARP_CYLINDRICAL


def _named_warper(name):
    """A RotationWarper subclass for one of the reference's --warp names (CLI:556-570), e.g. PaniniWarper(scale, 2, 1)
    == WARPERS["paniniA2B1"](scale)."""
    return type(name[0].upper() + name[1:].replace(".", "_") + "Warper", (RotationWarper,), {"kind": capi.WARP_NAMES[name]})


WARPERS = {"plane": PlaneWarper, "spherical": SphericalWarper, "cylindrical": CylindricalWarper}
for _n in capi.WARP_NAMES:
    WARPERS.setdefault(_n, _named_warper(_n))
WARP_KINDS = dict(capi.WARP_NAMES)
BLEND_KINDS = {"no": capi.BLEND_NO, "feather": capi.BLEND_FEATHER, "multiband": capi.BLEND_MULTI_BAND}
EXPOS_KINDS = {"no": capi.EXPOS_NO, "gain": capi.EXPOS_GAIN, "gain_blocks": capi.EXPOS_GAIN_BLOCKS}


class VoronoiSeamFinder:
    """cv::detail::VoronoiSeamFinder (seam_finders.hpp:97-104): find(sizes, corners, masks) updates the masks in place."""

    def __init__(self, device=0):
        self.device = device

    def find(self, sizes, corners, masks):
        n = len(masks)
        for m, (w, h) in zip(masks, sizes):
            if m.dtype != np.uint8 or m.shape != (h, w) or not m.flags.c_contiguous or not m.flags.writeable:
                raise capi.StitchError(capi.ERR_INVALID_ARG, "masks must be writable C-contiguous CV_8U of the given sizes")
        cxy = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
        swh = (C.c_int * (2 * n))(*[int(v) for s in sizes for v in s])
        mp = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
        mstr = (C.c_size_t * n)(*[m.strides[0] for m in masks])
        capi.check(capi.load().mst_seam_voronoi(self.device, n, cxy, swh, mp, mstr))


class DpSeamFinder:
    """cv::detail::DpSeamFinder (seam_finders.hpp:95-178): find(src, corners, masks) updates the masks in place.
    cost: "color" (DpSeamFinder::COLOR, --seam dp_color) or "color_grad" (--seam dp_colorgrad).  src: CV_8UC3 or CV_32FC3."""

    def __init__(self, cost="color", device=0):
        self.cost = {"color": 0, "color_grad": 1}[cost]
        self.device = device

    def find(self, src, corners, masks):
        n = len(masks)
        is_f32 = src[0].dtype != np.uint8
        imgs = [np.ascontiguousarray(im, np.float32 if is_f32 else np.uint8) for im in src]
        for im, m in zip(imgs, masks):
            if im.ndim != 3 or im.shape[2] != 3 or m.dtype != np.uint8 or m.shape != im.shape[:2] or not m.flags.c_contiguous or \
                    not m.flags.writeable:
                raise capi.StitchError(capi.ERR_INVALID_ARG, "src must be CV_8UC3 / CV_32FC3, masks writable C-contiguous CV_8U of the same sizes")
        cxy = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
        swh = (C.c_int * (2 * n))(*[v for im in imgs for v in (im.shape[1], im.shape[0])])
        ip = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        istr = (C.c_size_t * n)(*[im.strides[0] for im in imgs])
        mp = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
        mstr = (C.c_size_t * n)(*[m.strides[0] for m in masks])
        capi.check(capi.load().mst_seam_dp(self.device, n, cxy, swh, ip, istr, 1 if is_f32 else 0, mp, mstr, self.cost))


def _feed(device, kind, corners, images, masks, block=(32, 32)):
    """ExposureCompensator::feed (exposure_compensate.cpp:63-147,165-222) -> gains (list) or gain maps (list of arrays)."""
    n = len(images)
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    msk = [np.ascontiguousarray(m, np.uint8) for m in masks]
    for im, m in zip(imgs, msk):
        if im.ndim != 3 or im.shape[2] != 3 or m.shape != im.shape[:2]:
            raise capi.StitchError(capi.ERR_INVALID_ARG, "feed expects CV_8UC3 images and CV_8U masks of the same size")
    L = capi.load()
    cxy = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
    swh = (C.c_int * (2 * n))(*[v for im in imgs for v in (im.shape[1], im.shape[0])])
    ip = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
    istr = (C.c_size_t * n)(*[im.strides[0] for im in imgs])
    mp = (C.c_void_p * n)(*[m.ctypes.data for m in msk])
    mstr = (C.c_size_t * n)(*[m.strides[0] for m in msk])
    if kind == capi.EXPOS_GAIN:
        gains = (C.c_double * n)()
        capi.check(L.mst_exposure_feed(device, kind, n, cxy, ip, istr, swh, mp, mstr, 0, 0, gains, None))
        return list(gains)
    maps = []
    for im in imgs:
        gw, gh = C.c_int(), C.c_int()
        capi.check(L.mst_gain_map_size(im.shape[1], im.shape[0], block[0], block[1], C.byref(gw), C.byref(gh)))
        maps.append(np.empty((gh.value, gw.value), np.float32))
    gp = (C.c_void_p * n)(*[g.ctypes.data for g in maps])
    capi.check(L.mst_exposure_feed(device, kind, n, cxy, ip, istr, swh, mp, mstr, block[0], block[1], None, gp))
    return maps


class GainCompensator:
    """feed(): gains from the overlaps (exposure_compensate.cpp:73-147); apply(): image *= gain (:150-153)."""

    def __init__(self, gains=(), device=0):
        self.gains = list(gains)
        self.device = device

    def feed(self, corners, images, masks):
        self.gains = _feed(self.device, capi.EXPOS_GAIN, corners, images, masks)

    def apply(self, index, corner, image, mask=None):
        _exposure_apply(self.device, capi.EXPOS_GAIN, image, None, float(self.gains[index]))


class BlocksGainCompensator:
    """feed(): per-block gains, smoothed twice (exposure_compensate.cpp:165-222); apply(): bilinearly up-sized block
    gain map (:225-246)."""

    def __init__(self, gain_maps=(), device=0, bl_width=32, bl_height=32):
        self.gain_maps = [np.ascontiguousarray(g, np.float32) for g in gain_maps]
        self.device = device
        self.block = (bl_width, bl_height)

    def feed(self, corners, images, masks):
        self.gain_maps = _feed(self.device, capi.EXPOS_GAIN_BLOCKS, corners, images, masks, self.block)

    def apply(self, index, corner, image, mask=None):
        _exposure_apply(self.device, capi.EXPOS_GAIN_BLOCKS, image, self.gain_maps[index], 1.0)


def _exposure_apply(device, kind, image, gain_map, gain):
    if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3 or image.strides[2] != 1 \
            or image.strides[1] != 3:
        raise capi.StitchError(capi.ERR_INVALID_ARG, "apply expects a CV_8UC3 image")   # CV_Assert :227
    gp, gw, gh = None, 0, 0
    if gain_map is not None:
        gp, gw, gh = gain_map.ctypes.data, gain_map.shape[1], gain_map.shape[0]
    capi.check(capi.load().mst_exposure_apply(device, kind, image.ctypes.data, image.shape[1], image.shape[0],
                                              image.strides[0], gp, gw, gh, gain))


def multiband_geometry(canvas, num_bands, image):
    """MultiBandBlender::prepare/feed integer geometry.  -> (padded_canvas, used_bands, out[8])"""
    pad, nb, out = capi.Rect(), C.c_int(), (C.c_int * 8)()
    capi.check(capi.load().mst_multiband_geometry(capi.Rect(*canvas), num_bands, capi.Rect(*image),
                                                  C.byref(pad), C.byref(nb), out))
    return pad.tuple(), nb.value, list(out)


def init_undistort_rectify_map(K, dist, size, R=None, newK=None):
    """cv::initUndistortRectifyMap(K, dist, R, newK, size, CV_16SC2) (DRV:232) ->
    (map1 int16 (h, w, 2), map2 uint16 (h, w)); host only."""
    w, h = size
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    newK = K if newK is None else np.ascontiguousarray(newK, np.float64).reshape(9)
    dist = np.ascontiguousarray(dist, np.float64).reshape(-1)
    Rp = None if R is None else np.ascontiguousarray(R, np.float64).reshape(9)
    m1 = np.empty((h, w, 2), np.int16)
    m2 = np.empty((h, w), np.uint16)
    capi.check(capi.load().mst_init_undistort_rectify_map(
        K.ctypes.data, dist.ctypes.data if dist.size else None, dist.size, None if Rp is None else Rp.ctypes.data,
        newK.ctypes.data, w, h, m1.ctypes.data, m2.ctypes.data))
    return m1, m2


def cvt_nv12_bgr(yuv, device=0):
    """cv::cvtColor(yuv, COLOR_YUV2BGR_NV12) on the device: (H * 3 / 2, W) uint8 -> (H, W, 3) BGR."""
    yuv = np.ascontiguousarray(yuv, np.uint8)
    if yuv.ndim != 2 or yuv.shape[0] % 3:
        raise ValueError("NV12 frames are (H * 3 / 2, W) uint8")
    h, w = yuv.shape[0] * 2 // 3, yuv.shape[1]
    out = np.empty((h, w, 3), np.uint8)
    capi.check(capi.load().mst_cvt_nv12_bgr(device, yuv.ctypes.data, w, h, yuv.strides[0], out.ctypes.data, out.strides[0]))
    return out


def compose_lens_maps(xmap, ymap, map1, map2):
    """The map composition of PanoramaEngine(lens_composed=True) (mst_compose_lens_maps): the lens maps interpolated at the
    warp's backward map; returns new (xmap, ymap).  Host only."""
    x = np.ascontiguousarray(xmap, np.float32).copy()
    y = np.ascontiguousarray(ymap, np.float32).copy()
    m1 = np.ascontiguousarray(map1, np.int16)
    m2 = np.ascontiguousarray(map2, np.uint16)
    h, w = m2.shape
    if m1.shape != (h, w, 2):
        raise ValueError("map1 must be (h, w, 2) int16 and map2 (h, w) uint16")
    capi.check(capi.load().mst_compose_lens_maps(x.ctypes.data, y.ctypes.data, x.size, m1.ctypes.data, m2.ctypes.data, w, h))
    return x, y


def remap_fixed(image, map1, map2, device=0):
    """cv::remap(image, dst, map1, map2, INTER_LINEAR) with CV_16SC2 / CV_16UC1 maps on a CV_8UC3 image (DRV:741)."""
    image = np.ascontiguousarray(image, np.uint8)
    h, w = image.shape[:2]
    m1 = np.ascontiguousarray(map1, np.int16)
    m2 = np.ascontiguousarray(map2, np.uint16)
    if m1.shape != (h, w, 2) or m2.shape != (h, w):
        raise ValueError("maps must have the image's size")
    out = np.empty_like(image)
    capi.check(capi.load().mst_remap_fixed(device, image.ctypes.data, w, h, image.strides[0], m1.ctypes.data,
                                           m2.ctypes.data, out.ctypes.data, out.strides[0]))
    return out


class PinnedArray:
    """A numpy view of page-locked host memory from mst_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype=np.uint8):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._L = capi.load()
        self._p = self._L.mst_host_alloc(self.nbytes)
        if not self._p:
            raise capi.StitchError(capi.ERR_OUT_OF_MEMORY, "mst_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self._p)).view(dtype).reshape(shape)

    def __del__(self):
        if getattr(self, "_p", None):
            self._L.mst_host_free(self._p)
            self._p = None


def synth_frame(camera, frame_index, width, height):
    """Synthetic BGR frame of SURVEY.md section 8d (texture + PCG32 noise)."""
    out = np.empty((height, width, 3), np.uint8)
    capi.check(capi.load().mst_synth_frame(camera, frame_index, width, height, out.ctypes.data, out.strides[0]))
    return out


def calibration_struct(scale, cameras):
    """(scale, list of camera dicts) -> (capi.Calibration, the arrays it points into: keep them alive while it is used).
    Camera dict keys: size (w, h), K, R, [T], [gain_map | gain], mask | seam_mask, [undistort = (map1, map2)]."""
    n = len(cameras)
    arr = (capi.Camera * n)()
    keep = []
    for i, c in enumerate(cameras):
        arr[i].src_width, arr[i].src_height = int(c["size"][0]), int(c["size"][1])
        arr[i].K = _f9(c["K"])
        arr[i].R = _f9(c["R"])
        arr[i].T = _f3(c.get("T"))
        g = c.get("gain_map")
        if g is not None:
            g = np.ascontiguousarray(g, np.float32)
            keep.append(g)
            arr[i].gain_map, arr[i].gain_w, arr[i].gain_h = g.ctypes.data, g.shape[1], g.shape[0]
        arr[i].gain = float(c.get("gain", 1.0))
        sm = c.get("seam_mask")                # seam-scale mask: the engine does the fix-up (stitcher.cpp:291-294)
        if sm is not None:
            sm = np.ascontiguousarray(sm, np.uint8)
            keep.append(sm)
            arr[i].seam_mask, arr[i].seam_mask_width, arr[i].seam_mask_height = sm.ctypes.data, sm.shape[1], sm.shape[0]
            arr[i].seam_mask_stride = sm.strides[0]
        if c.get("mask") is not None:
            m = np.ascontiguousarray(c["mask"], np.uint8)
            keep.append(m)
            arr[i].mask, arr[i].mask_width, arr[i].mask_height = m.ctypes.data, m.shape[1], m.shape[0]
            arr[i].mask_stride = m.strides[0]
        und = c.get("undistort")               # (map1 int16 (h, w, 2), map2 uint16 (h, w)) or None
        if und is not None:
            m1 = np.ascontiguousarray(und[0], np.int16)
            m2 = np.ascontiguousarray(und[1], np.uint16)
            if m1.shape != (arr[i].src_height, arr[i].src_width, 2) or m2.shape != m1.shape[:2]:
                raise ValueError("undistort maps must be frame-sized (h, w, 2) int16 and (h, w) uint16")
            keep += [m1, m2]
            arr[i].undistort_map1, arr[i].undistort_map2 = m1.ctypes.data, m2.ctypes.data
    cal = capi.Calibration(C.sizeof(capi.Calibration), n, float(scale), arr)
    keep.append(arr)
    return cal, keep


def save_calibration(scale, cameras):
    """The calibration as one self-contained binary (mst_calibration_save) -> bytes."""
    cal, keep = calibration_struct(scale, cameras)
    L = capi.load()
    size = L.mst_calibration_blob_size(C.byref(cal))
    if not size:
        raise ValueError("not a valid calibration")
    buf = (C.c_uint8 * size)()
    written = C.c_size_t()
    capi.check(L.mst_calibration_save(C.byref(cal), buf, size, C.byref(written)))
    return bytes(buf[:written.value])


def load_calibration(blob):
    """bytes from save_calibration -> (scale, list of camera dicts as PanoramaEngine.update takes them)."""
    L = capi.load()
    raw = np.frombuffer(bytes(blob), np.uint8).copy()
    cams = (capi.Camera * capi.MAX_CAMERAS)()
    cal = capi.Calibration()
    capi.check(L.mst_calibration_load(raw.ctypes.data, raw.size, C.byref(cal), cams, capi.MAX_CAMERAS))

    def arr(ptr, shape, dtype):
        if not ptr:
            return None
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        off = ptr - raw.ctypes.data
        return raw[off:off + n].view(dtype).reshape(shape).copy()

    out = []
    for i in range(cal.num_cameras):
        c = cams[i]
        d = dict(size=(c.src_width, c.src_height), K=np.array(c.K[:], np.float32).reshape(3, 3),
                 R=np.array(c.R[:], np.float32).reshape(3, 3), T=np.array(c.T[:], np.float32), gain=float(c.gain))
        g = arr(c.gain_map, (c.gain_h, c.gain_w), np.float32)
        if g is not None:
            d["gain_map"] = g
        m = arr(c.mask, (c.mask_height, c.mask_width), np.uint8)
        if m is not None:
            d["mask"] = m
        sm = arr(c.seam_mask, (c.seam_mask_height, c.seam_mask_width), np.uint8)
        if sm is not None:
            d["seam_mask"] = sm
        if c.undistort_map1:
            d["undistort"] = (arr(c.undistort_map1, (c.src_height, c.src_width, 2), np.int16),
                              arr(c.undistort_map2, (c.src_height, c.src_width), np.uint16))
        out.append(d)
    return float(cal.warped_image_scale), out


class PanoramaEngine:
    """N frames in, one panorama out, every frame (the reference's StitchingAll loop).

    cameras passed to update(): list of dicts with
      K (3x3), R (3x3), optional T (3), size (width, height), mask (roi-sized uint8),
      optional gain_map (2-D float32) / gain (float).
    """

    def __init__(self, num_cameras, warp="spherical", blend="multiband", expos="gain_blocks", num_bands=5,
                 feather_sharpness=0.02, device=0, driver_lut=False, crop=(0.1, 0.1, 10.0, 10.0), debug=False,
                 overlap_frames=False, max_batch=0, weight_type="32f", warp_ab=None, strip=None, lens_composed=False, frames_nv12=False, graph_replay=False, overlap_depth=0):
        self._L = capi.load()
        self._h = C.c_void_p()
        cfg = capi.Config()
        cfg.struct_size = C.sizeof(capi.Config)
        cfg.num_cameras = num_cameras
        cfg.warp_kind = WARP_KINDS[warp] if isinstance(warp, str) else warp
        cfg.blend_kind = BLEND_KINDS[blend] if isinstance(blend, str) else blend
        cfg.expos_kind = EXPOS_KINDS[expos] if isinstance(expos, str) else expos
        cfg.num_bands = num_bands
        cfg.feather_sharpness = feather_sharpness
        cfg.device = device
        cfg.flags = (capi.FLAG_DRIVER_LUT if driver_lut else 0) | (capi.FLAG_DEBUG if debug else 0) | \
            (capi.FLAG_OVERLAP_FRAMES if overlap_frames else 0) | (capi.FLAG_LENS_COMPOSED if lens_composed else 0) | (capi.FLAG_FRAMES_NV12 if frames_nv12 else 0) | \
            (capi.FLAG_GRAPH_REPLAY if graph_replay else 0)
        self._frames_nv12 = bool(frames_nv12)
        cfg.upblack, cfg.downblack, cfg.leftblack, cfg.rightblack = crop
        cfg.max_batch = max_batch
        cfg.weight_type = {"32f": 0, "16s": 1}[weight_type] if isinstance(weight_type, str) else int(weight_type)
        if warp_ab is not None:          # CompressedRectilinear* / Panini* with arbitrary (A, B)
            cfg.warp_kind = (cfg.warp_kind & 0xFF) | capi.WARP_AB_CUSTOM
            cfg.warp_a, cfg.warp_b = float(warp_ab[0]), float(warp_ab[1])
        cfg.overlap_depth = int(overlap_depth)         # with overlap_frames: calls in flight (3..8: whole calls on own streams)
        if strip is not None:            # (index, count): this engine composes one column strip of the canvas (canvas split over GPUs)
            cfg.strip_index, cfg.strip_count = int(strip[0]), int(strip[1])
        self.num_cameras = num_cameras
        self.device = device
        capi.check(self._L.mst_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.mst_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        capi.check(rc, self._h)

    def update(self, scale, cameras):
        cal, keep = calibration_struct(scale, cameras)
        self._check(self._L.mst_update(self._h, C.byref(cal)))

    def acquire(self):
        """Latch the newest calibration for the following compose calls (the frame thread's
        `if (Upflag) UpdateMat();`, DRV:835-836) -> (canvas rect, generation)."""
        r, g = capi.Rect(), C.c_uint64()
        self._check(self._L.mst_acquire_calibration(self._h, C.byref(r), C.byref(g)))
        self._latching = True
        return r.tuple(), g.value

    def release(self):
        """Stop latching: every call picks up the newest published calibration again."""
        self._check(self._L.mst_release_calibration(self._h))
        self._latching = False

    def canvas(self):
        r = capi.Rect()
        self._check(self._L.mst_canvas(self._h, C.byref(r)))
        return r.tuple()

    def strip(self):
        """The columns this engine writes, as a rectangle inside canvas() (the whole canvas without strip=...)."""
        r = capi.Rect()
        self._check(self._L.mst_strip(self._h, C.byref(r)))
        return r.tuple()

    def camera_roi(self, i):
        r = capi.Rect()
        self._check(self._L.mst_camera_roi(self._h, i, C.byref(r)))
        return r.tuple()

    def camera_mask(self, i):
        """Final mask_warped of camera i as the engine holds it (the caller's, or the device-side fix-up result)."""
        _, _, w, h = self.camera_roi(i)
        m = np.empty((h, w), np.uint8)
        self._check(self._L.mst_camera_mask(self._h, i, m.ctypes.data, m.strides[0]))
        return m

    def num_bands(self):
        v = C.c_int()
        self._check(self._L.mst_num_bands(self._h, C.byref(v)))
        return v.value

    def compose(self, frames, want_mask=True):
        """frames: list of (H, W, 3) uint8 host arrays -> (pano uint8 (H, W, 3), mask uint8 or None)."""
        n = len(frames)
        if getattr(self, "_frames_nv12", False):      # (H * 3 / 2, W) uint8: Y rows, then interleaved U V rows
            fr = [f if (f.dtype == np.uint8 and f.ndim == 2 and f.strides[1] == 1) else np.ascontiguousarray(f, np.uint8) for f in frames]
        else:
            fr = [f if (f.dtype == np.uint8 and f.strides[2] == 1 and f.strides[1] == 3) else np.ascontiguousarray(f, np.uint8)
                  for f in frames]
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
        strides = (C.c_size_t * n)(*[f.strides[0] for f in fr])
        # latch first, size the buffers from the latched canvas, compose against the same calibration: a concurrent
        # update() cannot change the panorama size in between.  An engine whose user never asked for latching goes
        # back to "newest calibration on entry" afterwards.
        was_latching = getattr(self, "_latching", False)
        r, g = capi.Rect(), C.c_uint64()
        self._check(self._L.mst_acquire_calibration(self._h, C.byref(r), C.byref(g)))
        try:
            _, _, w, h = r.tuple()
            pano = np.empty((h, w, 3), np.uint8)
            mask = np.empty((h, w), np.uint8) if want_mask else None
            self._check(self._L.mst_compose(self._h, ptrs, strides, pano.ctypes.data, pano.strides[0],
                                            mask.ctypes.data if want_mask else None, w))
        finally:
            if not was_latching:
                self._L.mst_release_calibration(self._h)
        return pano, mask

    def pipeline_submit(self, frames, pano):
        """Queue one frame set of the three-stage host pipeline (frames / pano: uint8 arrays, ideally PinnedArray
        views); returns at once.  Keep the arrays alive and untouched until the matching pipeline_wait()."""
        n = len(frames)
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        strides = (C.c_size_t * n)(*[f.strides[0] for f in frames])
        self._check(self._L.mst_pipeline_submit(self._h, ptrs, strides, pano.ctypes.data, pano.strides[0]))

    def pipeline_submit_shm(self, clients, pano, timeout_ms=20000):
        """Queue one frame set pulled straight from the capture processes' shared-memory segments (mst_pipeline_submit_shm:
        clients = one ingest.FrameClient per camera) into the host pipeline; `pano`: page-locked uint8 array for the panorama."""
        n = len(clients)
        L = self._L
        if not getattr(L, "_submit_shm_ready", False):
            L.mst_pipeline_submit_shm.restype = C.c_int
            L.mst_pipeline_submit_shm.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_int]
            L._submit_shm_ready = True
        hs = (C.c_void_p * n)(*[c._h for c in clients])
        self._check(L.mst_pipeline_submit_shm(self._h, hs, n, pano.ctypes.data, pano.strides[0], timeout_ms))

    def pipeline_wait(self):
        """Block until the oldest submitted frame is done -> (address of its panorama buffer, canvas rect)."""
        p, r = C.c_void_p(), capi.Rect()
        self._check(self._L.mst_pipeline_wait(self._h, C.byref(p), C.byref(r)))
        return p.value, r.tuple()

    def compose_device(self, frame_ptrs, frame_strides, pano_ptr, pano_stride, mask_ptr=None, mask_stride=0,
                       stream=None):
        """Device pointers (ints) already resident in HBM; asynchronous on `stream`."""
        n = len(frame_ptrs)
        ptrs = (C.c_void_p * n)(*frame_ptrs)
        strides = (C.c_size_t * n)(*frame_strides)
        self._check(self._L.mst_compose_device(self._h, ptrs, strides, pano_ptr, pano_stride, mask_ptr,
                                               mask_stride, stream))

    def make_device_call(self, frame_ptrs, frame_strides, pano_ptr, pano_stride, mask_ptr=None, mask_stride=0,
                         stream=None):
        """Pre-marshal a compose_device call; returns a zero-argument callable (for tight frame loops)."""
        n = len(frame_ptrs)
        ptrs = (C.c_void_p * n)(*frame_ptrs)
        strides = (C.c_size_t * n)(*frame_strides)
        f, h = self._L.mst_compose_device, self._h
        pano_ptr, mask_ptr, stream = C.c_void_p(pano_ptr), C.c_void_p(mask_ptr), C.c_void_p(stream)

        def call():
            rc = f(h, ptrs, strides, pano_ptr, pano_stride, mask_ptr, mask_stride, stream)
            if rc:
                self._check(rc)
        return call

    def _batch_tables(self, frame_ptrs, frame_strides, pano_ptrs, mask_ptrs):
        count = len(pano_ptrs)
        flat_p = [p for fs in frame_ptrs for p in fs]
        flat_s = [s for fs in frame_strides for s in fs]
        if len(flat_p) != count * self.num_cameras or len(flat_s) != len(flat_p):
            raise ValueError("need num_cameras frame pointers and strides per frame set")
        ptrs = (C.c_void_p * len(flat_p))(*flat_p)
        strides = (C.c_size_t * len(flat_s))(*flat_s)
        panos = (C.c_void_p * count)(*pano_ptrs)
        masks = (C.c_void_p * count)(*mask_ptrs) if mask_ptrs else None
        return count, ptrs, strides, panos, masks

    def compose_device_batch(self, frame_ptrs, frame_strides, pano_ptrs, pano_stride, mask_ptrs=None, mask_stride=0,
                             stream=None):
        """len(pano_ptrs) frame sets (lists of per-camera device pointers / strides) -> that many panoramas,
        one launch sequence (needs max_batch >= the count at construction)."""
        count, ptrs, strides, panos, masks = self._batch_tables(frame_ptrs, frame_strides, pano_ptrs, mask_ptrs)
        self._check(self._L.mst_compose_device_batch(self._h, count, ptrs, strides, panos, pano_stride, masks,
                                                     mask_stride, stream))

    def make_device_batch_call(self, frame_ptrs, frame_strides, pano_ptrs, pano_stride, mask_ptrs=None, mask_stride=0,
                               stream=None):
        """Pre-marshalled compose_device_batch; returns a zero-argument callable (for tight frame loops)."""
        count, ptrs, strides, panos, masks = self._batch_tables(frame_ptrs, frame_strides, pano_ptrs, mask_ptrs)
        f, h, stream = self._L.mst_compose_device_batch, self._h, C.c_void_p(stream)

        def call():
            rc = f(h, count, ptrs, strides, panos, pano_stride, masks, mask_stride, stream)
            if rc:
                self._check(rc)
        return call

    def compose_device_batch_ex(self, frame_ptrs, frame_strides, pano_ptrs, pano_stride, mask_ptrs=None, mask_stride=0,
                                stream=None, frames_ready=None, outputs_free=None):
        """compose_device_batch behind explicit dependencies (mst_compose_device_batch_ex): `frames_ready` / `outputs_free` are raw
        hipEvent_t handles (torch.cuda.Event().cuda_event after record()) the call's kernels wait for -- the producer's event
        behind the last write of these frames, the consumer's behind the last read of these panorama buffers.  With calls in
        flight (overlap_depth >= 3) nothing else orders a call behind work queued on other streams."""
        count, ptrs, strides, panos, masks = self._batch_tables(frame_ptrs, frame_strides, pano_ptrs, mask_ptrs)
        self._check(self._L.mst_compose_device_batch_ex(self._h, count, ptrs, strides, panos, pano_stride, masks,
                                                        mask_stride, stream, frames_ready, outputs_free))

    def profile_stages_batch(self, frame_ptrs, frame_strides, pano_ptrs, pano_stride, stream=None, max_stages=64):
        """One batched launch sequence with a HIP event between consecutive launches -> [(kernel name, ms), ...]."""
        count, ptrs, strides, panos, _ = self._batch_tables(frame_ptrs, frame_strides, pano_ptrs, None)
        names = ((C.c_char * 32) * max_stages)()
        ms = (C.c_float * max_stages)()
        num = C.c_int()
        self._check(self._L.mst_compose_profile_batch(self._h, count, ptrs, strides, panos, pano_stride, stream,
                                                      max_stages, C.cast(names, C.c_void_p), ms, C.byref(num)))
        k = min(num.value, max_stages)
        return [(names[i].value.decode(), float(ms[i])) for i in range(k)]

    # ---- camera streams sharded over GPUs: per-camera stage, pyramid hand-off, canvas stage ----
    def stage_cameras(self, cameras, frame_ptrs, frame_strides, stream=None):
        """Run the per-camera kernels for `cameras` (iterable of indices); frame tables are indexed by camera."""
        n = self.num_cameras
        mask = 0
        for c in cameras:
            mask |= 1 << c
        ptrs = (C.c_void_p * n)(*[frame_ptrs.get(i) if isinstance(frame_ptrs, dict) else frame_ptrs[i] for i in range(n)])
        strides = (C.c_size_t * n)(*[(frame_strides.get(i, 0) if isinstance(frame_strides, dict) else frame_strides[i])
                                     for i in range(n)])
        self._check(self._L.mst_stage_cameras(self._h, mask, ptrs, strides, stream))

    def stage_canvas(self, pano_ptr, pano_stride, mask_ptr=None, mask_stride=0, stream=None):
        self._check(self._L.mst_stage_canvas(self._h, pano_ptr, pano_stride, mask_ptr, mask_stride, stream))

    def payload_bytes(self, camera):
        """Bytes of the camera's hand-off payload (launched tile interiors of every level, packed)."""
        b = C.c_size_t()
        self._check(self._L.mst_payload_bytes(self._h, camera, C.byref(b)))
        return b.value

    def payload_pack(self, camera, buf_ptr, stream=None):
        self._check(self._L.mst_payload_pack(self._h, camera, buf_ptr, stream))

    def payload_unpack(self, camera, buf_ptr, stream=None):
        self._check(self._L.mst_payload_unpack(self._h, camera, buf_ptr, stream))

    def select_stage_set(self, s):
        self._check(self._L.mst_stage_select_set(self._h, s))

    def pyramid_buffer(self, camera):
        """-> (device pointer, bytes) of the camera's contiguous Gaussian pyramid block."""
        p, b = C.c_void_p(), C.c_size_t()
        self._check(self._L.mst_pyramid_buffer(self._h, camera, C.byref(p), C.byref(b)))
        return p.value, b.value

    def profile_stages(self, frame_ptrs, frame_strides, pano_ptr, pano_stride, stream=None, max_stages=64):
        """One frame with a HIP event between consecutive launches -> [(kernel name, ms), ...]."""
        n = len(frame_ptrs)
        ptrs = (C.c_void_p * n)(*frame_ptrs)
        strides = (C.c_size_t * n)(*frame_strides)
        names = ((C.c_char * 32) * max_stages)()
        ms = (C.c_float * max_stages)()
        count = C.c_int()
        self._check(self._L.mst_compose_profile(self._h, ptrs, strides, pano_ptr, pano_stride, stream, max_stages,
                                                C.cast(names, C.c_void_p), ms, C.byref(count)))
        k = min(count.value, max_stages)
        return [(names[i].value.decode(), float(ms[i])) for i in range(k)]

    def synchronize(self):
        self._check(self._L.mst_synchronize(self._h))

    def debug_read(self, what, camera=0, level=0):
        w, h = C.c_int(), C.c_int()
        self._check(self._L.mst_debug_read(self._h, what, camera, level, None, 0, C.byref(w), C.byref(h)))
        if what in (capi.DBG_CAM_WEIGHT_F32, capi.DBG_BAND_WEIGHT_F32):
            buf = np.empty((h.value, w.value), np.float32)
        elif what == capi.DBG_CAM_NEED_U8:
            buf = np.empty((h.value, w.value), np.uint8)
        else:
            buf = np.empty((3, h.value, w.value), np.int16)
        self._check(self._L.mst_debug_read(self._h, what, camera, level, buf.ctypes.data, buf.nbytes,
                                           C.byref(w), C.byref(h)))
        return buf

"""ctypes binding of the CPU oracle (oracle/build/libmst_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  The product package (stitchingvideo_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "build", "libmst_oracle.so")

WARP_PLANE, WARP_SPHERICAL, WARP_CYLINDRICAL = 0, 1, 2
# MO_WARP_* of mst_oracle.h: the same numbering, and the same A/B presets in bits 8-9, as the product's MST_WARP_*
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
BLEND_NO, BLEND_FEATHER, BLEND_MULTI_BAND = 0, 1, 2
EXPOS_NO, EXPOS_GAIN, EXPOS_GAIN_BLOCKS = 0, 1, 2
WEIGHT_32F, WEIGHT_16S = 0, 1


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]

    def tuple(self):
        return (self.x, self.y, self.width, self.height)


class Projector(C.Structure):
    _fields_ = [("scale", C.c_float), ("k", C.c_float * 9), ("rinv", C.c_float * 9),
                ("r_kinv", C.c_float * 9), ("k_rinv", C.c_float * 9), ("t", C.c_float * 3),
                ("a", C.c_float), ("b", C.c_float)]


class ComposeConfig(C.Structure):
    _fields_ = [("warp_kind", C.c_int), ("scale", C.c_float), ("blend_kind", C.c_int),
                ("num_bands", C.c_int), ("sharpness", C.c_float), ("weight_type", C.c_int),
                ("expos_kind", C.c_int), ("num_threads", C.c_int)]


class ComposeCamera(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("R", C.c_float * 9), ("T", C.c_float * 3),
                ("frame", C.c_void_p), ("src_w", C.c_int), ("src_h", C.c_int),
                ("src_stride", C.c_size_t),
                ("gain_map", C.c_void_p), ("gain_w", C.c_int), ("gain_h", C.c_int),
                ("gain", C.c_double), ("mask", C.c_void_p),
                ("lens_map1", C.c_void_p), ("lens_map2", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("MST_ORACLE_NATIVE") == "1":
        # bench.py's cpu_baseline leg: the -O3 -march=native build, compiled on the host that times it (never shipped)
        path = os.path.join(ORACLE_DIR, "build", "libmst_oracle_native.so")
        subprocess.check_call(["make", "-s", "-B", "-C", ORACLE_DIR, "native"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(path)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    vp = C.c_void_p
    L.mo_cv_round.restype = C.c_int; L.mo_cv_round.argtypes = [C.c_double]
    L.mo_trunc_s16.restype = C.c_int16; L.mo_trunc_s16.argtypes = [C.c_float]
    L.mo_border_interpolate.restype = C.c_int; L.mo_border_interpolate.argtypes = [C.c_int] * 3
    L.mo_set_num_threads.argtypes = [C.c_int]
    L.mo_set_camera_params.argtypes = [C.POINTER(Projector), fp, fp, fp]
    L.mo_map_forward.argtypes = [C.POINTER(Projector), C.c_int, C.c_float, C.c_float, fp, fp]
    L.mo_map_backward.argtypes = [C.POINTER(Projector), C.c_int, C.c_float, C.c_float, fp, fp]
    L.mo_warp_roi.restype = Rect
    L.mo_warp_roi.argtypes = [C.c_int, C.c_float, fp, fp, fp, C.c_int, C.c_int]
    L.mo_build_maps.restype = Rect
    L.mo_build_maps.argtypes = [C.c_int, C.c_float, fp, fp, fp, C.c_int, C.c_int, vp, vp]
    L.mo_remap_8u.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, C.c_int, C.c_int,
                              C.c_size_t, vp, vp, C.c_size_t, C.c_int, C.c_int]
    L.mo_warp_8u.restype = Rect
    L.mo_warp_8u.argtypes = [C.c_int, C.c_float, fp, fp, fp, vp, C.c_int, C.c_int, C.c_size_t,
                             C.c_int, C.c_int, C.c_int, vp]
    L.mo_cvt_nv12_bgr.restype = None
    L.mo_cvt_nv12_bgr.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]
    L.mo_compose_lens_maps.restype = None
    L.mo_compose_lens_maps.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_int, C.c_int]
    L.mo_init_undistort_rectify_map.restype = C.c_int
    L.mo_init_undistort_rectify_map.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    L.mo_remap_8u_fixed.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, C.c_int, C.c_int,
                                    C.c_size_t, vp, vp]
    L.mo_bilinear_tab_i.restype = C.POINTER(C.c_int16)
    L.mo_resize_linear_32f.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.mo_blocks_gain_apply.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int, C.c_int]
    L.mo_gain_apply.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double]
    L.mo_copy_make_border_16s.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp] + [C.c_int] * 5
    L.mo_copy_make_border_32f.argtypes = [vp, C.c_int, C.c_int, vp] + [C.c_int] * 5
    L.mo_pyr_down_16s.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.mo_pyr_down_32f.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.mo_pyr_up_16s.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.mo_add_16s.argtypes = [vp, vp, vp, C.c_size_t]
    L.mo_subtract_16s.argtypes = [vp, vp, vp, C.c_size_t]
    L.mo_distance_transform_l1.argtypes = [vp, C.c_int, C.c_int, vp]
    L.mo_result_roi.restype = Rect; L.mo_result_roi.argtypes = [ip, ip, C.c_int]
    L.mo_create_weight_map.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp]
    L.mo_normalize_using_weight_map_32f.argtypes = [vp, vp, C.c_int, C.c_int]
    L.mo_blender_create.restype = vp
    L.mo_blender_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int]
    L.mo_blender_destroy.argtypes = [vp]
    L.mo_blender_prepare.argtypes = [vp, ip, ip, C.c_int]
    L.mo_blender_prepare_rect.argtypes = [vp, Rect]
    L.mo_blender_feed.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mo_blender_blend.restype = Rect
    L.mo_blender_blend.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.mo_free.argtypes = [vp]
    L.mo_blender_num_bands.restype = C.c_int; L.mo_blender_num_bands.argtypes = [vp]
    L.mo_blender_dst_roi.restype = Rect; L.mo_blender_dst_roi.argtypes = [vp]
    L.mo_blender_band.restype = vp; L.mo_blender_band.argtypes = [vp, C.c_int, ip, ip]
    L.mo_blender_band_weight.restype = vp; L.mo_blender_band_weight.argtypes = [vp, C.c_int, ip, ip]
    L.mo_multiband_feed_geometry.argtypes = [Rect] + [C.c_int] * 5 + [ip]
    L.mo_compose_panorama.restype = Rect
    L.mo_compose_panorama.argtypes = [C.POINTER(ComposeConfig), C.POINTER(ComposeCamera), C.c_int,
                                      C.POINTER(vp), C.POINTER(vp)]
    L.mo_lut_build.restype = vp
    L.mo_lut_build.argtypes = [ip, ip, C.POINTER(vp), C.c_int]
    L.mo_lut_canvas.restype = Rect; L.mo_lut_canvas.argtypes = [vp]
    L.mo_lut_compose.restype = Rect
    L.mo_lut_compose.argtypes = [vp, C.POINTER(vp)] + [C.c_float] * 4 + [C.POINTER(vp)]
    L.mo_lut_destroy.argtypes = [vp]
    _lib = L
    return L


def _f9(a):
    return (C.c_float * 9)(*np.asarray(a, np.float32).reshape(-1))


def _f3(a):
    if a is None:
        a = (0, 0, 0)
    return (C.c_float * 3)(*np.asarray(a, np.float32).reshape(-1))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- thin numpy wrappers ---------------------------------------------------------------------

def warp_roi(kind, scale, K, R, src_w, src_h, T=None):
    r = lib().mo_warp_roi(kind, scale, _f9(K), _f9(R), _f3(T), src_w, src_h)
    return r.tuple()


def build_maps(kind, scale, K, R, src_w, src_h, T=None):
    x, y, w, h = warp_roi(kind, scale, K, R, src_w, src_h, T)
    xmap = np.empty((h, w), np.float32)
    ymap = np.empty((h, w), np.float32)
    lib().mo_build_maps(kind, scale, _f9(K), _f9(R), _f3(T), src_w, src_h, _ptr(xmap), _ptr(ymap))
    return (x, y, w, h), xmap, ymap


def remap(src, xmap, ymap, interp, border):
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    xmap = np.ascontiguousarray(xmap, np.float32)
    ymap = np.ascontiguousarray(ymap, np.float32)
    dh, dw = xmap.shape
    dst = np.empty((dh, dw) + (() if src.ndim == 2 else (cn,)), np.uint8)
    lib().mo_remap_8u(_ptr(src), sw, sh, sw * cn, cn, _ptr(dst), dw, dh, dw * cn,
                      _ptr(xmap), _ptr(ymap), dw, interp, border)
    return dst


def init_undistort_rectify_map(K, dist, size, R=None, newK=None):
    """initUndistortRectifyMap(K, dist, R, newK, size, CV_16SC2) -> (map1 int16 (h, w, 2), map2 uint16 (h, w))."""
    w, h = size
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    newK = K if newK is None else np.ascontiguousarray(newK, np.float64).reshape(9)
    dist = np.ascontiguousarray(dist, np.float64).reshape(-1)
    Rp = None if R is None else np.ascontiguousarray(R, np.float64).reshape(9)
    m1 = np.empty((h, w, 2), np.int16)
    m2 = np.empty((h, w), np.uint16)
    rc = lib().mo_init_undistort_rectify_map(_ptr(K), _ptr(dist), dist.size, None if Rp is None else _ptr(Rp),
                                             _ptr(newK), w, h, _ptr(m1), _ptr(m2))
    if rc != 0:
        raise ValueError("bad distortion coefficient count")
    return m1, m2


def compose_lens_maps(xmap, ymap, map1, map2):
    """The fused lens variant's map composition (mst_oracle.h); returns new (xmap, ymap)."""
    x = np.ascontiguousarray(xmap, np.float32).copy()
    y = np.ascontiguousarray(ymap, np.float32).copy()
    m1 = np.ascontiguousarray(map1, np.int16)
    m2 = np.ascontiguousarray(map2, np.uint16)
    h, w = m2.shape
    lib().mo_compose_lens_maps(_ptr(x), _ptr(y), C.c_size_t(x.size), _ptr(m1), _ptr(m2), w, h)
    return x, y


def cvt_nv12_bgr(yuv):
    """cvtColor(COLOR_YUV2BGR_NV12): yuv (h * 3 / 2, w) uint8 -> (h, w, 3) BGR."""
    yuv = np.ascontiguousarray(yuv, np.uint8)
    h, w = yuv.shape[0] * 2 // 3, yuv.shape[1]
    out = np.empty((h, w, 3), np.uint8)
    lib().mo_cvt_nv12_bgr(_ptr(yuv), w, h, yuv.strides[0], _ptr(out), out.strides[0])
    return out


def remap_fixed(src, map1, map2):
    """remap(src, dst, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR) (BORDER_CONSTANT 0)."""
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    map1 = np.ascontiguousarray(map1, np.int16)
    map2 = np.ascontiguousarray(map2, np.uint16)
    dh, dw = map2.shape
    dst = np.empty((dh, dw) + (() if src.ndim == 2 else (cn,)), np.uint8)
    lib().mo_remap_8u_fixed(_ptr(src), sw, sh, sw * cn, cn, _ptr(dst), dw, dh, dw * cn, _ptr(map1), _ptr(map2))
    return dst


def warp(kind, scale, K, R, src, interp, border, T=None):
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    x, y, w, h = warp_roi(kind, scale, K, R, sw, sh, T)
    dst = np.empty((h, w) + (() if src.ndim == 2 else (cn,)), np.uint8)
    lib().mo_warp_8u(kind, scale, _f9(K), _f9(R), _f3(T), _ptr(src), sw, sh, sw * cn, cn,
                     interp, border, _ptr(dst))
    return (x, y, w, h), dst


def resize_linear_32f(src, dw, dh):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.empty((dh, dw), np.float32)
    lib().mo_resize_linear_32f(_ptr(src), src.shape[1], src.shape[0], _ptr(dst), dw, dh)
    return dst


def blocks_gain_apply(img, gain_map):
    img = np.ascontiguousarray(img).copy()
    g = np.ascontiguousarray(gain_map, np.float32)
    lib().mo_blocks_gain_apply(_ptr(img), img.shape[1], img.shape[0], img.shape[1] * 3, _ptr(g),
                               g.shape[1], g.shape[0])
    return img


def gain_apply(img, gain):
    img = np.ascontiguousarray(img).copy()
    cn = 1 if img.ndim == 2 else img.shape[2]
    lib().mo_gain_apply(_ptr(img), img.shape[1], img.shape[0], img.shape[1] * cn, cn, float(gain))
    return img


def pyr_down_16s(src):
    src = np.ascontiguousarray(src, np.int16)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty(((h + 1) // 2, (w + 1) // 2) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().mo_pyr_down_16s(_ptr(src), w, h, cn, _ptr(dst))
    return dst


def pyr_down_32f(src):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape[:2]
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().mo_pyr_down_32f(_ptr(src), w, h, 1, _ptr(dst))
    return dst


def pyr_up_16s(src):
    src = np.ascontiguousarray(src, np.int16)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((2 * h, 2 * w) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().mo_pyr_up_16s(_ptr(src), w, h, cn, _ptr(dst), 2 * w, 2 * h)
    return dst


def copy_make_border_16s(src, top, bottom, left, right, border):
    src = np.ascontiguousarray(src, np.int16)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((h + top + bottom, w + left + right) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().mo_copy_make_border_16s(_ptr(src), w, h, cn, _ptr(dst), top, bottom, left, right, border)
    return dst


def distance_transform_l1(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    dst = np.empty(mask.shape, np.float32)
    lib().mo_distance_transform_l1(_ptr(mask), mask.shape[1], mask.shape[0], _ptr(dst))
    return dst


def create_weight_map(mask, sharpness):
    mask = np.ascontiguousarray(mask, np.uint8)
    dst = np.empty(mask.shape, np.float32)
    lib().mo_create_weight_map(_ptr(mask), mask.shape[1], mask.shape[0], sharpness, _ptr(dst))
    return dst


def _feed_args(corners, images, masks):
    n = len(images)
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    msk = [np.ascontiguousarray(m, np.uint8) for m in masks]
    cxy = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
    swh = (C.c_int * (2 * n))(*[v for im in imgs for v in (im.shape[1], im.shape[0])])
    ip = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
    istr = (C.c_size_t * n)(*[im.strides[0] for im in imgs])
    mp = (C.c_void_p * n)(*[m.ctypes.data for m in msk])
    mstr = (C.c_size_t * n)(*[m.strides[0] for m in msk])
    return n, imgs, msk, cxy, swh, ip, istr, mp, mstr


def gain_feed(corners, images, masks):
    """GainCompensator::feed -> gains (list of float)."""
    n, imgs, msk, cxy, swh, ip, istr, mp, mstr = _feed_args(corners, images, masks)
    g = (C.c_double * n)()
    lib().mo_gain_feed(n, cxy, ip, istr, swh, mp, mstr, g)
    return list(g)


def blocks_gain_feed(corners, images, masks, block=(32, 32)):
    """BlocksGainCompensator::feed -> list of float32 gain maps."""
    n, imgs, msk, cxy, swh, ip, istr, mp, mstr = _feed_args(corners, images, masks)
    maps = []
    for im in imgs:
        gw, gh = C.c_int(), C.c_int()
        lib().mo_gain_map_size(im.shape[1], im.shape[0], block[0], block[1], C.byref(gw), C.byref(gh))
        maps.append(np.empty((gh.value, gw.value), np.float32))
    gp = (C.c_void_p * n)(*[g.ctypes.data for g in maps])
    lib().mo_blocks_gain_feed(n, cxy, ip, istr, swh, mp, mstr, block[0], block[1], gp)
    return maps


def solve_lu(A, b):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(len(b), np.float64)
    ok = lib().mo_solve_lu_64f(_ptr(A), _ptr(b), len(b), _ptr(x))
    return ok, x


def voronoi_seams(corners, masks):
    """VoronoiSeamFinder::find -> new list of masks."""
    n = len(masks)
    out = [np.ascontiguousarray(m, np.uint8).copy() for m in masks]
    cxy = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
    swh = (C.c_int * (2 * n))(*[v for m in out for v in (m.shape[1], m.shape[0])])
    mp = (C.c_void_p * n)(*[m.ctypes.data for m in out])
    lib().mo_voronoi_seams(n, cxy, swh, mp)
    return out


def dp_seams(corners, images, masks, cost="color"):
    """DpSeamFinder(COLOR | COLOR_GRAD)::find -> new list of masks.  images: CV_32FC3 or CV_8UC3 (h, w, 3) arrays."""
    n = len(masks)
    out = [np.ascontiguousarray(m, np.uint8).copy() for m in masks]
    is_float = images[0].dtype != np.uint8
    imgs = [np.ascontiguousarray(im, np.float32 if is_float else np.uint8) for im in images]
    cxy = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
    swh = (C.c_int * (2 * n))(*[v for m in out for v in (m.shape[1], m.shape[0])])
    ip = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
    mp = (C.c_void_p * n)(*[m.ctypes.data for m in out])
    lib().mo_dp_seams(n, cxy, swh, ip, 1 if is_float else 0, {"color": 0, "color_grad": 1}[cost], mp)
    return out


def dp_gradients(image):
    """DpSeamFinder::computeGradients of one CV_32FC3 / CV_8UC3 image -> (gx, gy) float32 (h, w)."""
    is_float = image.dtype != np.uint8
    im = np.ascontiguousarray(image, np.float32 if is_float else np.uint8)
    h, w = im.shape[:2]
    gx, gy = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    f = lib().mo_dp_gradients
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f(im.ctypes.data, 1 if is_float else 0, w, h, gx.ctypes.data, gy.ctypes.data)
    return gx, gy


def dilate3x3(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    dst = np.empty_like(mask)
    lib().mo_dilate3x3_8u(_ptr(mask), mask.shape[1], mask.shape[0], _ptr(dst))
    return dst


def resize_linear_8u(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty((dh, dw), np.uint8)
    lib().mo_resize_linear_8u(_ptr(src), src.shape[1], src.shape[0], _ptr(dst), dw, dh)
    return dst


def resize_linear_8u_factor(src, f):
    """cv::resize(src, dst, Size(), f, f, INTER_LINEAR) on CV_8UC1 / CV_8UC3 (the sample's --compose_megapix resize, CLI:699-700)"""
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = int(np.rint(sw * f)), int(np.rint(sh * f))
    dst = np.zeros((dh, dw) + src.shape[2:], np.uint8)
    lib().mo_resize_linear_8u_cn(_ptr(src), sw, sh, cn, _ptr(dst), dw, dh, C.c_double(1.0 / f), C.c_double(1.0 / f))
    return dst


def mask_fixup(seam_mask, mask_warped):
    """stitcher.cpp:291-294: dilate -> resize to the warped size -> & the warped all-255 mask."""
    seam_mask = np.ascontiguousarray(seam_mask, np.uint8)
    mask_warped = np.ascontiguousarray(mask_warped, np.uint8)
    out = np.empty_like(mask_warped)
    lib().mo_mask_fixup(_ptr(seam_mask), seam_mask.shape[1], seam_mask.shape[0], _ptr(mask_warped),
                        mask_warped.shape[1], mask_warped.shape[0], _ptr(out))
    return out


def multiband_feed_geometry(dst_roi, num_bands, tl, size):
    out = (C.c_int * 8)()
    lib().mo_multiband_feed_geometry(Rect(*dst_roi), num_bands, tl[0], tl[1], size[0], size[1], out)
    return list(out)


class Blender:
    """Mirror of cv::detail::Blender / FeatherBlender / MultiBandBlender (oracle side)."""

    def __init__(self, kind, num_bands=5, sharpness=0.02, weight_type=WEIGHT_32F):
        self.L = lib()
        self.kind = kind
        self.h = self.L.mo_blender_create(kind, num_bands, sharpness, weight_type)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mo_blender_destroy(self.h)
            self.h = None

    def prepare(self, corners, sizes):
        n = len(corners)
        c = (C.c_int * (2 * n))(*[v for p in corners for v in p])
        s = (C.c_int * (2 * n))(*[v for p in sizes for v in p])
        self.L.mo_blender_prepare(self.h, c, s, n)

    def feed(self, img16s, mask, tl):
        img16s = np.ascontiguousarray(img16s, np.int16)
        mask = np.ascontiguousarray(mask, np.uint8)
        h, w = mask.shape
        assert img16s.shape == (h, w, 3)
        self.L.mo_blender_feed(self.h, _ptr(img16s), _ptr(mask), w, h, tl[0], tl[1])

    def num_bands(self):
        return self.L.mo_blender_num_bands(self.h)

    def dst_roi(self):
        return self.L.mo_blender_dst_roi(self.h).tuple()

    def band(self, level):
        w, h = C.c_int(), C.c_int()
        p = self.L.mo_blender_band(self.h, level, C.byref(w), C.byref(h))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), (h.value, w.value, 3))
        return a.copy()

    def band_weight(self, level):
        w, h = C.c_int(), C.c_int()
        p = self.L.mo_blender_band_weight(self.h, level, C.byref(w), C.byref(h))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (h.value, w.value))
        return a.copy()

    def blend(self):
        d, m = C.c_void_p(), C.c_void_p()
        r = self.L.mo_blender_blend(self.h, C.byref(d), C.byref(m))
        dst = np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_int16)), (r.height, r.width, 3)).copy()
        msk = np.ctypeslib.as_array(C.cast(m, C.POINTER(C.c_uint8)), (r.height, r.width)).copy()
        self.L.mo_free(d); self.L.mo_free(m)
        return r.tuple(), dst, msk


def compose_panorama(warp_kind, scale, blend_kind, expos_kind, cams, num_bands=5, sharpness=0.02,
                     weight_type=WEIGHT_32F, num_threads=1):
    """cams: list of dicts with K, R, (T), frame (HxWx3 u8), mask (roi-sized u8),
    optional gain_map (float32 2-D) or gain (float).  Returns (rect, pano u8, mask u8)."""
    L = lib()
    cfg = ComposeConfig(warp_kind, scale, blend_kind, num_bands, sharpness, weight_type, expos_kind,
                        num_threads)
    arr = (ComposeCamera * len(cams))()
    keep = []
    for i, c in enumerate(cams):
        frame = np.ascontiguousarray(c["frame"], np.uint8)
        mask = np.ascontiguousarray(c["mask"], np.uint8)
        keep += [frame, mask]
        arr[i].K = _f9(c["K"]); arr[i].R = _f9(c["R"]); arr[i].T = _f3(c.get("T"))
        arr[i].frame = frame.ctypes.data
        arr[i].src_w = frame.shape[1]; arr[i].src_h = frame.shape[0]
        arr[i].src_stride = frame.strides[0]
        if c.get("gain_map") is not None:
            g = np.ascontiguousarray(c["gain_map"], np.float32)
            keep.append(g)
            arr[i].gain_map = g.ctypes.data; arr[i].gain_w = g.shape[1]; arr[i].gain_h = g.shape[0]
        arr[i].gain = float(c.get("gain", 1.0))
        arr[i].mask = mask.ctypes.data
        if c.get("lens_maps") is not None:          # the fused single-resampling variant: frame is the RAW frame
            m1 = np.ascontiguousarray(c["lens_maps"][0], np.int16)
            m2 = np.ascontiguousarray(c["lens_maps"][1], np.uint16)
            keep += [m1, m2]
            arr[i].lens_map1 = m1.ctypes.data; arr[i].lens_map2 = m2.ctypes.data
    p, m = C.c_void_p(), C.c_void_p()
    r = L.mo_compose_panorama(C.byref(cfg), arr, len(cams), C.byref(p), C.byref(m))
    pano = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (r.height, r.width, 3)).copy()
    pmask = np.ctypeslib.as_array(C.cast(m, C.POINTER(C.c_uint8)), (r.height, r.width)).copy()
    L.mo_free(p); L.mo_free(m)
    return r.tuple(), pano, pmask


class Lut:
    """The driver's LUT composite (DRV:94-177)."""

    def __init__(self, corners, sizes, masks):
        self.L = lib()
        n = len(corners)
        c = (C.c_int * (2 * n))(*[v for p in corners for v in p])
        s = (C.c_int * (2 * n))(*[v for p in sizes for v in p])
        self.masks = [np.ascontiguousarray(m, np.uint8) for m in masks]
        mp = (C.c_void_p * n)(*[m.ctypes.data for m in self.masks])
        self.h = self.L.mo_lut_build(c, s, mp, n)
        self.n = n

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mo_lut_destroy(self.h)
            self.h = None

    def canvas(self):
        return self.L.mo_lut_canvas(self.h).tuple()

    def compose(self, imgs16s, up=0.1, down=0.1, left=10.0, right=10.0):
        imgs = [np.ascontiguousarray(i, np.int16) for i in imgs16s]
        ip = (C.c_void_p * self.n)(*[i.ctypes.data for i in imgs])
        p = C.c_void_p()
        r = self.L.mo_lut_compose(self.h, ip, up, down, left, right, C.byref(p))
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (r.height, r.width, 3)).copy()
        self.L.mo_free(p)
        return r.tuple(), out

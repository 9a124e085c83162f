"""TEST INFRASTRUCTURE, OPT-IN: the reference's own shipped x64 machine code, run in place on numpy buffers (oracle/ref_run/pe_run.c maps
the images and calls functions by address).  Nothing here runs unless MST_RUN_REFERENCE_BINARIES=1 is set AND /root/reference holds
the four binaries with the SHA-256 pinned below; the default build and test paths only read the committed vectors these runs produced
(tests/golden/ref_runs.npz, tests/golden/ref_blend.npz).

Every run happens in a short-lived child process -- `python tests/ref_run.py <job> <in.npz> <out.npz> <scratch>` is that process,
`run(job, **arrays)` the caller's side -- which, after mapping the images and before the first instruction of theirs executes, confines
ITSELF (pe_lockdown: uid/gid nobody when it was root, an empty scratch directory, no_new_privs, a seccomp filter on all threads that
leaves memory management, reads/writes on descriptors already open, clocks and exit).  pe_call refuses to run in an unconfined process.
The parent reads the answer as plain arrays (np.load, allow_pickle off).  cv::Mat headers are laid out by hand (OpenCV 2.4.11, x64: 96
bytes)."""
import ctypes as C
import hashlib
import io
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RELEASE = "/root/reference/stitching/OpenCV2.4.11-Stitching-64bit/x64/Release"
DLLS = {"core": "opencv_core2411.dll", "imgproc": "opencv_imgproc2411.dll", "stitching": "opencv_stitching2411.dll",
        "exe": "OpenCV2.4.11-Stitching.exe"}                    # (the x64 executable maps at 0x140000000: the reference's OWN compiled sources)
PINNED = {"core": "cfedf4b256c68aa18a89ce1f7f75d4e3fdaa6fd721b3133b63577919be5c1e60",
          "imgproc": "5f0902d99da96d26db4bb304f047365eb7e308af53da1f47bd0c0907d1344a0f",
          "stitching": "0d9365a326abaa4d8cd3a0e0ed6b80d3848927fff24ff146d0c56b6e3c0d09c3",
          "exe": "6ea54dc72eefb94e8f2681ec31bac41f731c7e7dca44de49348a4e2aea2f6997"}      # the files the audit read; anything else is not run
OPT_IN = "MST_RUN_REFERENCE_BINARIES"
SRC = os.path.join(ROOT, "oracle", "ref_run", "pe_run.c")
LIB = os.path.join(ROOT, "oracle", "ref_run", "libpe_run.so")
PE_LEAF, PE_FULL = 0, 1
SCRATCH = None                  # the empty directory the child confines itself to (argv[4] of the job process)


def opted_in():
    return os.environ.get(OPT_IN) == "1"


def available():
    """the reference binaries may be run here: opted in, present, and byte for byte the pinned files"""
    return opted_in() and all(os.path.exists(os.path.join(RELEASE, d)) for d in DLLS.values())


def image_bytes(which):
    data = open(os.path.join(RELEASE, DLLS[which]), "rb").read()
    sha = hashlib.sha256(data).hexdigest()
    if sha != PINNED[which]:
        raise RuntimeError("%s is not the pinned file (sha256 %s): refusing to run it" % (DLLS[which], sha))
    return data


def build():
    if not opted_in():
        raise RuntimeError("running the reference's binaries is opt-in: set %s=1" % OPT_IN)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


class Mat(C.Structure):              # cv::Mat of OpenCV 2.4.11 on x64
    _fields_ = [("flags", C.c_int), ("dims", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("data", C.c_void_p), ("refcount", C.c_void_p),
                ("datastart", C.c_void_p), ("dataend", C.c_void_p), ("datalimit", C.c_void_p), ("allocator", C.c_void_p),
                ("size_p", C.c_void_p), ("step_p", C.c_void_p), ("step_buf", C.c_size_t * 2)]


DEPTH = {np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3, np.dtype(np.int32): 4,
         np.dtype(np.float32): 5, np.dtype(np.float64): 6}


def mat(a):
    """a cv::Mat header over a C-contiguous numpy array (h, w) or (h, w, cn); the array must outlive it"""
    assert a.flags["C_CONTIGUOUS"] and a.ndim in (2, 3)
    cn = a.shape[2] if a.ndim == 3 else 1
    m = Mat()
    m.flags = 0x42FF0000 | (1 << 14) | (DEPTH[a.dtype] + ((cn - 1) << 3))
    m.dims, m.rows, m.cols = 2, a.shape[0], a.shape[1]
    m.data = m.datastart = a.ctypes.data
    m.dataend = m.datalimit = a.ctypes.data + a.nbytes
    m.refcount = m.allocator = None
    m.size_p = C.addressof(m) + Mat.rows.offset
    m.step_p = C.addressof(m) + Mat.step_buf.offset
    m.step_buf[0], m.step_buf[1] = a.shape[1] * cn * a.itemsize, cn * a.itemsize
    return m


_L = None


def _lib():
    global _L
    if _L is not None:
        return _L
    L = _L = C.CDLL(build())
    L.pe_map.restype = C.c_int
    L.pe_map.argtypes = [C.c_char_p, C.c_long, C.c_uint64, C.c_char_p]
    L.pe_bind.restype = C.c_int
    L.pe_bind.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pe_export.restype = C.c_uint64
    L.pe_export.argtypes = [C.c_int, C.c_char_p]
    L.pe_base.restype = L.pe_entry.restype = C.c_uint64
    L.pe_base.argtypes = L.pe_entry.argtypes = [C.c_int]
    L.pe_lockdown.restype = C.c_int
    L.pe_lockdown.argtypes = [C.c_char_p]
    L.pe_call.restype = C.c_uint64
    L.pe_call.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    L.pe_allow_libm.restype = C.c_int
    L.pe_allow_libm.argtypes = [C.c_int]
    L.pe_phase.restype = C.c_int
    L.pe_phase.argtypes = [C.c_int]
    L.pe_math_log.restype = C.c_int
    L.pe_math_log.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return L


def _confine(L):
    """no instruction of the reference executes before this returns 0"""
    if SCRATCH is None:
        raise RuntimeError("reference code runs only in the job process (tests/ref_run.py <job> ...), which is given a scratch directory")
    np.savez(io.BytesIO(), warm=np.zeros(1))                    # (pull in every lazy import the answer needs while files can still be opened)
    rc = L.pe_lockdown(SCRATCH.encode())
    if rc != 0:
        raise RuntimeError("pe_lockdown failed (%d): not running reference code unconfined" % rc)


class Image:
    """ONE image at its preferred base, every import trapped but cv::checkHardwareSupport and memset / memcpy / memmove: leaf functions"""

    def __init__(self, which):
        self.L = _lib()
        assert self.L.pe_prepare() == 0
        data = image_bytes(which)
        idx = self.L.pe_map(data, len(data), 0, DLLS[which].encode())
        assert idx == 0, "the image did not map"
        self.base = self.L.pe_base(idx)
        assert self.base == (0x140000000 if which == "exe" else 0x180000000), "the image did not map at its preferred base"
        assert self.L.pe_bind(idx, PE_LEAF, None, None) >= 0
        _confine(self.L)

    def call(self, addr, *args):
        a = (C.c_uint64 * 8)(*([int(x) & 0xFFFFFFFFFFFFFFFF for x in args] + [0] * (8 - len(args))))
        return self.L.pe_call(addr, a)


class Process:
    """core + imgproc + stitching mapped TOGETHER (stitching at the shared preferred base, the other two relocated), imports bound to each
    other's exports by name, loader / allocator services answered by the host (pe_run.c: PE_FULL), every other import trapped by name.
    The images' own entry points run first, as the Windows loader would run them (_DllMainCRTStartup: the static initialisers -- cpuid
    feature flags, the MatOp singletons, the lookup tables)."""
    BASES = {"stitching": 0, "core": 0x190000000, "imgproc": 0x1A0000000, "exe": 0}      # (the exe's preferred base is 0x140000000: free)
    ORDER = ("core", "imgproc", "stitching")                    # dependency order: the loader initialises imports first

    def __init__(self, images=ORDER):
        self.L = _lib()
        assert self.L.pe_prepare() == 0
        self.idx, self.base = {}, {}
        for which in images:
            data = image_bytes(which)
            i = self.L.pe_map(data, len(data), self.BASES[which], DLLS[which].encode())
            assert i >= 0, "%s did not map" % which
            self.idx[which], self.base[which] = i, self.L.pe_base(i)
        self.binding, self.keep = {}, []
        for which in images:
            nb, na = C.c_int(), C.c_int()
            trapped = self.L.pe_bind(self.idx[which], PE_FULL, C.byref(nb), C.byref(na))
            assert trapped >= 0
            self.binding[which] = {"bound_to_images": nb.value, "answered_by_host": na.value, "trapped": trapped}
        _confine(self.L)
        for which in images:                                    # DllMain(hinst, DLL_PROCESS_ATTACH, NULL) must return TRUE
            if which == "exe":                                  # (its entry point is the program: not run -- only functions of its own are called)
                continue
            assert self.call(self.L.pe_entry(self.idx[which]), self.base[which], 1, 0) & 0xFF == 1, which
        self.trig_calls_in_initialisers = self.L.pe_phase(1)    # from here on sin / cos are trapped again
        # cv::setNumThreads(0): cv::parallel_for_ then runs its body over the whole range on the calling thread instead of handing it
        # to the Concurrency runtime (MSVCR100's, not in the tree).  Row-parallel loops: the results do not depend on the split.
        if "core" in self.idx:
            self.call(self.sym("core", "?setNumThreads@cv@@YAXH@Z"), 0)

    def sym(self, which, name):
        a = self.L.pe_export(self.idx[which], name.encode())
        assert a, "%s does not export %s" % (which, name)
        return a

    def at(self, which, preferred_address):
        """an address quoted for the image at its preferred base -> where it is now"""
        return preferred_address - (0x140000000 if which == "exe" else 0x180000000) + self.base[which]

    def call(self, addr, *args):
        a = (C.c_uint64 * 8)(*([int(x) & 0xFFFFFFFFFFFFFFFF for x in args] + [0] * (8 - len(args))))
        return self.L.pe_call(addr, a)

    # ---- whole exported functions of the reference's core / imgproc (cv::Mat in, a copy of the cv::Mat it allocated out) ----------
    def ia(self, m):
        """cv::_InputArray(const Mat&) / _OutputArray(Mat&), built by the reference's own constructors"""
        buf = (C.c_uint8 * 64)()
        self.call(self.sym("core", "??0_InputArray@cv@@QEAA@AEBVMat@1@@Z"), C.addressof(buf), C.addressof(m))
        self.keep += [buf, m]
        return C.addressof(buf)

    def oa(self, m):
        buf = (C.c_uint8 * 64)()
        self.call(self.sym("core", "??0_OutputArray@cv@@QEAA@AEAVMat@1@@Z"), C.addressof(buf), C.addressof(m))
        self.keep += [buf, m]
        return C.addressof(buf)

    def cv(self, which, name, *args):
        return self.call(self.sym(which, name), *args)

    def remap(self, src, xmap, ymap, interp, border):
        """cv::remap(src, dst, xmap CV_32F, ymap CV_32F, interp, border, Scalar()) -- imgwarp.cpp of 2.4.11, WHOLE: the float -> fixed map
        conversion, the tiling, the table lookup and the border handling included"""
        s, mx, my = np.ascontiguousarray(src), np.ascontiguousarray(xmap, np.float32), np.ascontiguousarray(ymap, np.float32)
        dst, zero = empty_mat(), (C.c_double * 4)()
        self.cv("imgproc", "?remap@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@00HHAEBV?$Scalar_@N@1@@Z", self.ia(mat(s)), self.oa(dst),
                self.ia(mat(mx)), self.ia(mat(my)), int(interp), int(border), C.addressof(zero))
        self.keep += [s, mx, my]
        return mat_array(dst)

    def resize(self, src, dsize, interp=1, fx=0.0, fy=0.0):
        """cv::resize(src, dst, dsize, fx, fy, interp), whole (the coefficient tables it builds included); dsize (0, 0) + factors is the
        sample's `resize(full_img, img, Size(), compose_scale, compose_scale)`"""
        s, dst, size = np.ascontiguousarray(src), empty_mat(), (C.c_int * 2)(int(dsize[0]), int(dsize[1]))
        bits = lambda v: int(np.float64(v).view(np.uint64))                    # noqa: E731
        self.cv("imgproc", "?resize@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@V?$Size_@H@1@NNH@Z", self.ia(mat(s)), self.oa(dst),
                C.addressof(size), bits(fx), bits(fy), int(interp))
        self.keep += [s]
        return mat_array(dst)

    def dilate(self, src):
        """cv::dilate(src, dst, Mat()) with every default (3 x 3 box, anchor (-1, -1), 1 iteration, BORDER_CONSTANT, morphologyDefaultBorderValue)"""
        s, dst, kernel, anchor = np.ascontiguousarray(src), empty_mat(), empty_mat(), (C.c_int * 2)(-1, -1)
        bv = (C.c_double * 4)(*([np.finfo(np.float64).max] * 4))
        self.cv("imgproc", "?dilate@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@0V?$Point_@H@1@HHAEBV?$Scalar_@N@1@@Z", self.ia(mat(s)),
                self.oa(dst), self.ia(kernel), C.addressof(anchor), 1, 0, C.addressof(bv))
        self.keep += [s]
        return mat_array(dst)

    def copy_make_border(self, src, top, bottom, left, right, border):
        s, dst, zero = np.ascontiguousarray(src), empty_mat(), (C.c_double * 4)()
        self.cv("imgproc", "?copyMakeBorder@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@HHHHHAEBV?$Scalar_@N@1@@Z", self.ia(mat(s)), self.oa(dst),
                int(top), int(bottom), int(left), int(right), int(border), C.addressof(zero))
        self.keep += [s]
        return mat_array(dst)

    def pyr_down(self, src):
        s, dst, size = np.ascontiguousarray(src), empty_mat(), (C.c_int * 2)(0, 0)
        self.cv("imgproc", "?pyrDown@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@AEBV?$Size_@H@1@H@Z", self.ia(mat(s)), self.oa(dst),
                C.addressof(size), BORDER_REFLECT_101)
        self.keep += [s]
        return mat_array(dst)

    def pyr_up(self, src, dsize):
        s, dst, size = np.ascontiguousarray(src), empty_mat(), (C.c_int * 2)(int(dsize[0]), int(dsize[1]))
        self.cv("imgproc", "?pyrUp@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@AEBV?$Size_@H@1@H@Z", self.ia(mat(s)), self.oa(dst),
                C.addressof(size), BORDER_REFLECT_101)
        self.keep += [s]
        return mat_array(dst)

    def convert_to(self, src, rtype, alpha=1.0, beta=0.0):
        """Mat::convertTo(dst, rtype, alpha, beta)"""
        s, dst = np.ascontiguousarray(src), empty_mat()
        ms = mat(s)
        bits = lambda v: int(np.float64(v).view(np.uint64))                    # noqa: E731
        self.cv("core", "?convertTo@Mat@cv@@QEBAXAEBV_OutputArray@2@HNN@Z", C.addressof(ms), self.oa(dst), int(rtype), bits(alpha), bits(beta))
        self.keep += [s, ms]
        return mat_array(dst)

    def bitwise_and(self, a, b):
        x, y, dst, nomask = np.ascontiguousarray(a), np.ascontiguousarray(b), empty_mat(), empty_mat()
        self.cv("core", "?bitwise_and@cv@@YAXAEBV_InputArray@1@0AEBV_OutputArray@1@0@Z", self.ia(mat(x)), self.ia(mat(y)), self.oa(dst), self.ia(nomask))
        self.keep += [x, y]
        return mat_array(dst)

    def distance_transform(self, mask, dist_type=1, mask_size=3):
        s, dst = np.ascontiguousarray(mask, np.uint8), empty_mat()
        self.cv("imgproc", "?distanceTransform@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@HH@Z", self.ia(mat(s)), self.oa(dst), int(dist_type),
                int(mask_size))
        self.keep += [s]
        return mat_array(dst)

    def math_log(self):
        out, fn, x, y = [], C.create_string_buffer(8), C.c_double(), C.c_double()
        n = self.L.pe_math_log(-1, fn, C.byref(x), C.byref(y))
        for i in range(n):
            self.L.pe_math_log(i, fn, C.byref(x), C.byref(y))
            out.append((fn.value.decode(), x.value, y.value))
        return out


def empty_mat():
    """a default-constructed cv::Mat (flags = MAGIC_VAL, dims 0, size.p -> rows, step.p -> step.buf)"""
    m = Mat()
    m.flags = 0x42FF0000
    m.size_p = C.addressof(m) + Mat.rows.offset
    m.step_p = C.addressof(m) + Mat.step_buf.offset
    return m


TYPES = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}


def mat_array(m):
    """a copy of what a cv::Mat the reference allocated holds"""
    depth, cn = m.flags & 7, ((m.flags >> 3) & 63) + 1
    dt = np.dtype(TYPES[depth])
    if not m.data or m.rows <= 0 or m.cols <= 0:
        return np.zeros((0, 0) if cn == 1 else (0, 0, cn), dt)
    step = C.c_size_t.from_address(m.step_p).value
    raw = np.frombuffer(C.string_at(m.data, step * (m.rows - 1) + m.cols * cn * dt.itemsize), np.uint8)
    rows = np.stack([raw[y * step:y * step + m.cols * cn * dt.itemsize] for y in range(m.rows)])
    out = rows.view(dt).reshape((m.rows, m.cols) if cn == 1 else (m.rows, m.cols, cn))
    return np.ascontiguousarray(out)


BLENDER = {"no": ("??0Blender@detail@cv@@QEAA@XZ", "Blender"), "feather": ("??0FeatherBlender@detail@cv@@QEAA@M@Z", "FeatherBlender"),
           "multiband": ("??0MultiBandBlender@detail@cv@@QEAA@HHH@Z", "MultiBandBlender")}


def job_blend(kind, corners, dst_roi, num_bands, weight_type, sharpness, **planes):
    """The reference's OWN blender, whole: cv::detail::{Blender, FeatherBlender, MultiBandBlender} of opencv_stitching2411.dll --
    constructor, prepare(Rect), feed(img CV_16SC3, mask CV_8U, tl) per camera, blend(dst, dst_mask) -- over core and imgproc mapped beside
    it, then `result.convertTo(pano, CV_8U)` as Stitcher::composePanorama / imwrite do (stitcher.cpp:320; DRV:787).  R64/src/blenders.cpp:
    71-113 (Blender), 115-186 (Feather), 189-377 (MultiBand), 435-530 (pyramids)."""
    kind = str(kind)
    p = Process()
    ctor, cls = BLENDER[kind]
    this = (C.c_uint8 * 4096)()
    obj = C.addressof(this)
    if kind == "multiband":
        p.call(p.sym("stitching", ctor), obj, 0, int(num_bands), int(weight_type))           # (try_gpu = false)
    elif kind == "feather":
        p.call(p.sym("stitching", ctor), obj, int(np.float32(sharpness).view(np.uint32)))
    else:
        p.call(p.sym("stitching", ctor), obj)
    rect = (C.c_int * 4)(*[int(v) for v in dst_roi])
    p.call(p.sym("stitching", "?prepare@%s@detail@cv@@UEAAXV?$Rect_@H@3@@Z" % cls), obj, C.addressof(rect))
    feed = p.sym("stitching", "?feed@%s@detail@cv@@UEAAXAEBVMat@3@0V?$Point_@H@3@@Z" % cls)
    keep = []
    for i, (x, y) in enumerate(np.asarray(corners).reshape(-1, 2)):
        img = np.ascontiguousarray(planes["img%d" % i], np.int16)
        msk = np.ascontiguousarray(planes["mask%d" % i], np.uint8)
        mi, mm = mat(img), mat(msk)
        keep += [img, msk, mi, mm]
        tl = (C.c_int * 2)(int(x), int(y))                      # cv::Point has a user-declared copy constructor: MSVC passes it by reference
        p.call(feed, obj, C.addressof(mi), C.addressof(mm), C.addressof(tl))
    dst, dst_mask = empty_mat(), empty_mat()
    p.call(p.sym("stitching", "?blend@%s@detail@cv@@UEAAXAEAVMat@3@0@Z" % cls), obj, C.addressof(dst), C.addressof(dst_mask))
    out = {"dst": mat_array(dst), "dst_mask": mat_array(dst_mask)}
    if kind == "multiband":
        out["num_bands"] = np.array(p.call(p.sym("stitching", "?numBands@MultiBandBlender@detail@cv@@QEBAHXZ"), obj) & 0xFFFFFFFF, np.int32)
    out["pano"] = p.convert_to(out["dst"], 0)                     # result.convertTo(pano, CV_8U)
    log = p.math_log()
    out["math_fn"] = np.array([l[0] for l in log], "S8")
    out["math_xy"] = np.array([[l[1], l[2]] for l in log], np.float64).reshape(-1, 2)
    print(p.binding, "trig calls in initialisers:", p.trig_calls_in_initialisers, file=sys.stderr)
    return out


def job_whole(fn, **a):
    """ONE whole exported function of the reference's imgproc / core (allocations and all), named by `fn`"""
    p, fn = Process(), str(fn)
    if fn == "remap":
        return {"out": p.remap(a["src"], a["xmap"], a["ymap"], int(a["interp"]), int(a["border"]))}
    if fn == "resize":
        return {"out": p.resize(a["src"], a["dsize"], int(a.get("interp", 1)))}
    if fn == "resize_factor":
        return {"out": p.resize(a["src"], (0, 0), 1, float(a["f"]), float(a["f"]))}
    if fn == "dilate":
        return {"out": p.dilate(a["src"])}
    if fn == "copy_make_border":
        return {"out": p.copy_make_border(a["src"], *[int(v) for v in a["tblr"]], int(a["border"]))}
    if fn == "pyr_down":
        return {"out": p.pyr_down(a["src"])}
    if fn == "pyr_up":
        return {"out": p.pyr_up(a["src"], a["dsize"])}
    if fn == "convert_to":
        return {"out": p.convert_to(a["src"], int(a["rtype"]), float(a["alpha"]), float(a["beta"]))}
    if fn == "distance_transform":
        return {"out": p.distance_transform(a["src"])}
    if fn == "init_undistort":
        # initUndistortRectifyMap(im1, D1, Mat(), im1, size, CV_16SC2, map1, map2) as RunTestEyeFishMap calls it (DRV:201-238)
        k = np.ascontiguousarray(a["K"], np.float64).reshape(3, 3)
        d = np.ascontiguousarray(a["dist"], np.float64).reshape(1, -1)
        m1, m2, no_r = empty_mat(), empty_mat(), empty_mat()
        size = (C.c_int * 2)(int(a["size"][0]), int(a["size"][1]))
        mk, md = mat(k), mat(d)
        p.cv("imgproc", "?initUndistortRectifyMap@cv@@YAXAEBV_InputArray@1@000V?$Size_@H@1@HAEBV_OutputArray@1@2@Z", p.ia(mk), p.ia(md), p.ia(no_r), p.ia(mk),
             C.addressof(size), 11, p.oa(m1), p.oa(m2))             # CV_16SC2 = 3 + (1 << 3)
        return {"map1": mat_array(m1), "map2": mat_array(m2)}
    if fn == "remap_fixed":
        # remap(Img, ImgUndistort, mapEye1, mapEye2, INTER_LINEAR): CV_16SC2 + CV_16UC1 maps straight in, BORDER_CONSTANT 0
        src, m1, m2 = np.ascontiguousarray(a["src"]), np.ascontiguousarray(a["map1"], np.int16), np.ascontiguousarray(a["map2"], np.uint16)
        dst, zero = empty_mat(), (C.c_double * 4)()
        p.cv("imgproc", "?remap@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@00HHAEBV?$Scalar_@N@1@@Z", p.ia(mat(src)), p.oa(dst), p.ia(mat(m1)), p.ia(mat(m2)),
             1, 0, C.addressof(zero))
        p.keep += [src, m1, m2]
        return {"out": mat_array(dst)}
    if fn == "nv12":
        src, dst = np.ascontiguousarray(a["src"], np.uint8), empty_mat()
        p.cv("imgproc", "?cvtColor@cv@@YAXAEBV_InputArray@1@AEBV_OutputArray@1@HH@Z", p.ia(mat(src)), p.oa(dst), 91, 0)       # COLOR_YUV2BGR_NV12
        p.keep += [src]
        return {"out": mat_array(dst)}
    if fn == "mask_fixup":
        # stitcher.cpp:291-293 / DRV:700-702: dilate(masks_warped[i], dilated, Mat()); resize(dilated, seam_mask, mask_warped.size());
        # mask_warped = seam_mask & mask_warped
        dil = p.dilate(a["seam_mask"])
        seam = p.resize(dil, (a["mask_warped"].shape[1], a["mask_warped"].shape[0]))
        return {"out": p.bitwise_and(seam, a["mask_warped"]), "dilated": dil, "resized": seam}
    raise KeyError(fn)


def job_pipeline(warp_rois, blend, num_bands, weight_type, sharpness, expos, **a):
    """The reference's compose loop for one frame set, every pixel operation by the reference's own code (stitcher.cpp:265-325 /
    DRV:697-790): per camera cv::remap(frame, xmap, ymap, INTER_LINEAR, BORDER_REFLECT) [whole], the exposure apply
    (BlocksGainCompensator::apply of the exe, cv::resize of the gain map inside it; or `img *= gain` = Mat::convertTo(-1, gain)),
    convertTo(CV_16S), Blender::feed; then blend and convertTo(CV_8U).  The float maps come from the caller (the projector's sinf / cosf
    are MSVCR100's, which the tree does not hold); the masks are the final seam masks of the rig."""
    blend, expos = str(blend), str(expos)
    rois = np.asarray(warp_rois).reshape(-1, 4)
    n = len(rois)
    p = Process(Process.ORDER + (("exe",) if expos == "gain_blocks" else ()))
    x0, y0 = rois[:, 0].min(), rois[:, 1].min()
    x1, y1 = (rois[:, 0] + rois[:, 2]).max(), (rois[:, 1] + rois[:, 3]).max()
    ctor, cls = BLENDER[blend]
    this = (C.c_uint8 * 4096)()
    obj = C.addressof(this)
    if blend == "multiband":
        p.call(p.sym("stitching", ctor), obj, 0, int(num_bands), int(weight_type))
    elif blend == "feather":
        p.call(p.sym("stitching", ctor), obj, int(np.float32(sharpness).view(np.uint32)))
    else:
        p.call(p.sym("stitching", ctor), obj)
    rect = (C.c_int * 4)(int(x0), int(y0), int(x1 - x0), int(y1 - y0))                     # resultRoi(corners, sizes): util.cpp:117-131
    out = {"rect": np.array(list(rect), np.int32)}              # (the callee owns the by-value Rect and pads it in place: read it first)
    p.call(p.sym("stitching", "?prepare@%s@detail@cv@@UEAAXV?$Rect_@H@3@@Z" % cls), obj, C.addressof(rect))
    feed = p.sym("stitching", "?feed@%s@detail@cv@@UEAAXAEBVMat@3@0V?$Point_@H@3@@Z" % cls)
    for i in range(n):
        warped = p.remap(a["frame%d" % i], a["xmap%d" % i], a["ymap%d" % i], 1, 2)
        mask = np.ascontiguousarray(a["mask%d" % i], np.uint8)
        assert warped.shape[:2] == mask.shape == (rois[i, 3], rois[i, 2])
        if expos == "gain_blocks":
            gm = np.ascontiguousarray(a["gain_map%d" % i], np.float32)
            mg, mi, mm = mat(gm), mat(warped), mat(mask)
            gthis = (C.c_uint64 * 8)()                                  # {vptr, (allocator), vector<Mat_<float>> gain_maps_ {first, last, end}}
            gthis[2] = C.addressof(mg)
            gthis[3] = gthis[4] = C.addressof(mg) + C.sizeof(Mat)
            corner = (C.c_int * 2)(int(rois[i, 0]), int(rois[i, 1]))
            p.call(p.at("exe", BLOCKS_GAIN_APPLY), C.addressof(gthis), 0, C.addressof(corner), C.addressof(mi), C.addressof(mm))
        elif expos == "gain":
            warped = p.convert_to(warped, -1, float(np.float64(a["gain%d" % i])))         # image *= gains_(index, 0): Mat::operator*=(double)
        if i == 0:
            out["warped0"] = warped.copy()
        img16 = p.convert_to(warped, 3)
        mi, mm, tl = mat(img16), mat(mask), (C.c_int * 2)(int(rois[i, 0]), int(rois[i, 1]))
        p.keep += [img16, mask, mi, mm, tl]
        p.call(feed, obj, C.addressof(mi), C.addressof(mm), C.addressof(tl))
    dst, dst_mask = empty_mat(), empty_mat()
    p.call(p.sym("stitching", "?blend@%s@detail@cv@@UEAAXAEAVMat@3@0@Z" % cls), obj, C.addressof(dst), C.addressof(dst_mask))
    out["pano"] = p.convert_to(mat_array(dst), 0)
    out["pano_mask"] = mat_array(dst_mask)
    log = p.math_log()
    out["math_fn"] = np.array([l[0] for l in log], "S8")
    out["math_xy"] = np.array([[l[1], l[2]] for l in log], np.float64).reshape(-1, 2)
    return out


def job_confinement_probe():
    """after the lockdown: an open, a socket and an exec must fail with EPERM, and the process must not be root"""
    import errno
    import socket
    Process(("core",))
    res = {}
    for name, fn in (("open", lambda: open("/etc/hostname")), ("socket", lambda: socket.socket()),
                     ("exec", lambda: os.execv("/bin/true", ["true"]))):
        try:
            fn()
            res[name + "_errno"] = np.array(0)
        except OSError as e:
            res[name + "_errno"] = np.array(e.errno if e.errno != errno.EACCES else 1)
    res["uid"] = np.array(os.getuid())
    return res


class _Vec(C.Structure):               # std::vector<T> of MSVC 2010, release build: _Myfirst, _Mylast, _Myend, _Alval (padded)
    _fields_ = [("first", C.c_void_p), ("last", C.c_void_p), ("end", C.c_void_p), ("alval", C.c_uint64)]


def _vec(buf, nbytes):
    v = _Vec()
    v.first = C.addressof(buf)
    v.last = v.end = C.addressof(buf) + nbytes
    return v


def _mat_into(slot_addr, a):
    """a cv::Mat header over array `a`, written AT slot_addr (its size / step pointers point into the slot)"""
    m = mat(a)
    C.memmove(slot_addr, C.addressof(m), C.sizeof(Mat))
    C.c_void_p.from_address(slot_addr + Mat.size_p.offset).value = slot_addr + Mat.rows.offset
    C.c_void_p.from_address(slot_addr + Mat.step_p.offset).value = slot_addr + Mat.step_buf.offset


def job_exposure_feed(kind, corners, block, **planes):
    """cv::detail::GainCompensator::feed / BlocksGainCompensator::feed of opencv_stitching2411.dll, WHOLE (exposure_compensate.cpp:70-142,
    172-222: overlap ROIs, the intensity sums, the normal equations, cv::solve(LU); for blocks: the per-block feed, the gain maps and their
    sepFilter2D smoothing) on std::vector<Point>, std::vector<Mat>, std::vector<pair<Mat, uchar>> laid out by hand."""
    kind = str(kind)
    p = Process()
    pts = np.ascontiguousarray(np.asarray(corners).reshape(-1, 2), np.int32)
    n = len(pts)
    imgs = [np.ascontiguousarray(planes["img%d" % i], np.uint8) for i in range(n)]
    masks = [np.ascontiguousarray(planes["mask%d" % i], np.uint8) for i in range(n)]
    v_images = (C.c_uint8 * (C.sizeof(Mat) * n))()
    v_masks = (C.c_uint8 * (104 * n))()                          # pair<Mat, uchar>: 96 + 1, padded to 104
    for i in range(n):
        _mat_into(C.addressof(v_images) + C.sizeof(Mat) * i, imgs[i])
        _mat_into(C.addressof(v_masks) + 104 * i, masks[i])
        C.c_uint8.from_address(C.addressof(v_masks) + 104 * i + 96).value = 255
    c_buf = (C.c_int32 * (2 * n))(*pts.ravel())
    vc, vi, vm = _vec(c_buf, 8 * n), _vec(v_images, C.sizeof(Mat) * n), _vec(v_masks, 104 * n)
    this = (C.c_uint8 * 1024)()
    obj = C.addressof(this)
    sig = ("@detail@cv@@UEAAXAEBV?$vector@V?$Point_@H@cv@@V?$allocator@V?$Point_@H@cv@@@std@@@std@@AEBV?$vector@VMat@cv@@V?$allocator@VMat@cv@@@std@@@5@"
           "AEBV?$vector@U?$pair@VMat@cv@@E@std@@V?$allocator@U?$pair@VMat@cv@@E@std@@@2@@5@@Z")
    if kind == "gain":
        p.call(p.sym("stitching", "??0GainCompensator@detail@cv@@QEAA@XZ"), obj)
        p.call(p.sym("stitching", "?feed@GainCompensator" + sig), obj, C.addressof(vc), C.addressof(vi), C.addressof(vm))
        gains = Mat.from_address(obj + 8)                        # {vptr, Mat_<double> gains_}
        return {"gains": mat_array(gains).reshape(-1)}
    bw, bh = int(block[0]), int(block[1])
    p.call(p.sym("stitching", "??0BlocksGainCompensator@detail@cv@@QEAA@HH@Z"), obj, bw, bh)
    p.call(p.sym("stitching", "?feed@BlocksGainCompensator" + sig), obj, C.addressof(vc), C.addressof(vi), C.addressof(vm))
    vec = _Vec.from_address(obj + 16)                            # {vptr, int bl_width_, bl_height_, vector<Mat_<float>> gain_maps_}
    cnt = (vec.last - vec.first) // C.sizeof(Mat)
    assert cnt == n, cnt
    return {"map%d" % i: mat_array(Mat.from_address(vec.first + C.sizeof(Mat) * i)) for i in range(n)}


def job_seams(kind, corners, **planes):
    """cv::detail::VoronoiSeamFinder / DpSeamFinder(COLOR | COLOR_GRAD)::find of opencv_stitching2411.dll, WHOLE (seam_finders.cpp:49-160,
    163-1045): images (CV_32FC3 as the sample passes them, or CV_8UC3), corners, masks modified in place."""
    kind = str(kind)
    p = Process()
    pts = np.ascontiguousarray(np.asarray(corners).reshape(-1, 2), np.int32)
    n = len(pts)
    imgs = [np.ascontiguousarray(planes["img%d" % i]) for i in range(n)]
    masks = [np.ascontiguousarray(planes["mask%d" % i], np.uint8).copy() for i in range(n)]
    v_images = (C.c_uint8 * (C.sizeof(Mat) * n))()
    v_masks = (C.c_uint8 * (C.sizeof(Mat) * n))()
    for i in range(n):
        _mat_into(C.addressof(v_images) + C.sizeof(Mat) * i, imgs[i])
        _mat_into(C.addressof(v_masks) + C.sizeof(Mat) * i, masks[i])
    c_buf = (C.c_int32 * (2 * n))(*pts.ravel())
    vc, vi, vm = _vec(c_buf, 8 * n), _vec(v_images, C.sizeof(Mat) * n), _vec(v_masks, C.sizeof(Mat) * n)
    this = (C.c_uint8 * 4096)()
    obj = C.addressof(this)
    sig = ("@detail@cv@@UEAAXAEBV?$vector@VMat@cv@@V?$allocator@VMat@cv@@@std@@@std@@AEBV?$vector@V?$Point_@H@cv@@V?$allocator@V?$Point_@H@cv@@@std@@@5@AEAV45@@Z")
    if kind == "voronoi":
        p.call(p.sym("stitching", "??0VoronoiSeamFinder@detail@cv@@QEAA@XZ"), obj)
        p.call(p.sym("stitching", "?find@PairwiseSeamFinder" + sig), obj, C.addressof(vi), C.addressof(vc), C.addressof(vm))
    else:
        p.call(p.sym("stitching", "??0DpSeamFinder@detail@cv@@QEAA@W4CostFunction@012@@Z"), obj, {"dp_color": 0, "dp_colorgrad": 1}[kind])
        p.call(p.sym("stitching", "?find@DpSeamFinder" + sig), obj, C.addressof(vi), C.addressof(vc), C.addressof(vm))
    return {"mask%d" % i: masks[i] for i in range(n)}


WARPERS = {"plane": ("PlaneWarper", "M"), "spherical": ("SphericalWarper", "M"), "cylindrical": ("CylindricalWarper", "M"),
           "fisheye": ("FisheyeWarper", "M"), "stereographic": ("StereographicWarper", "M"),
           "compressedPlane": ("CompressedRectilinearWarper", "MMM"), "compressedPlanePortrait": ("CompressedRectilinearPortraitWarper", "MMM"),
           "panini": ("PaniniWarper", "MMM"), "paniniPortrait": ("PaniniPortraitWarper", "MMM"), "mercator": ("MercatorWarper", "M"),
           "transverseMercator": ("TransverseMercatorWarper", "M"), "sphericalPortrait": ("SphericalPortraitWarper", "M"),
           "cylindricalPortrait": ("CylindricalPortraitWarper", "M"), "planePortrait": ("PlanePortraitWarper", "M")}
PROJECTOR = {"PlaneWarper": "PlaneProjector", "SphericalWarper": "SphericalProjector", "CylindricalWarper": "CylindricalProjector",
             "FisheyeWarper": "FisheyeProjector", "StereographicWarper": "StereographicProjector",
             "CompressedRectilinearWarper": "CompressedRectilinearProjector", "CompressedRectilinearPortraitWarper": "CompressedRectilinearPortraitProjector",
             "PaniniWarper": "PaniniProjector", "PaniniPortraitWarper": "PaniniPortraitProjector", "MercatorWarper": "MercatorProjector",
             "TransverseMercatorWarper": "TransverseMercatorProjector", "SphericalPortraitWarper": "SphericalPortraitProjector",
             "CylindricalPortraitWarper": "CylindricalPortraitProjector", "PlanePortraitWarper": "PlanePortraitProjector"}


def job_build_maps(warp, scale, ab, K, R, size):
    """cv::detail::<X>Warper::buildMaps(src_size, K, R, xmap, ymap) of opencv_stitching2411.dll, WHOLE (warpers.hpp / warpers_inl.hpp:
    setCameraParams, detectResultRoi[ByBorder], mapBackward per pixel): the float backward maps and the ROI.  The plane projector needs no
    math library; every other one calls MSVCR100's sinf / cosf / atan2f ..., which the tree does not hold: they are answered by THIS host's
    libm (pe_allow_libm), so such a run pins formulas, operation order and ROI detection, not the library's function values."""
    warp = str(warp)
    cls, sig = WARPERS[warp]
    p = Process()
    this = (C.c_uint8 * 512)()
    obj = C.addressof(this)
    f32 = lambda v: int(np.float32(v).view(np.uint32))          # noqa: E731
    ctor = p.sym("stitching", "??0%s@detail@cv@@QEAA@%s@Z" % (cls, sig))
    if sig == "M":
        p.call(ctor, obj, f32(scale))
    else:
        p.call(ctor, obj, f32(scale), f32(ab[0]), f32(ab[1]))
    k, r = np.ascontiguousarray(K, np.float32).reshape(3, 3), np.ascontiguousarray(R, np.float32).reshape(3, 3)
    mk, mr, xmap, ymap = mat(k), mat(r), empty_mat(), empty_mat()
    sz = (C.c_int * 2)(int(size[0]), int(size[1]))
    roi = (C.c_int * 4)()
    before = p.L.pe_allow_libm(0 if warp == "plane" else 1)
    if warp == "plane":                                          # PlaneWarper's own overload with a translation (zeros: what the virtual one passes)
        t = np.zeros((3, 1), np.float32)
        mt = mat(t)
        p.call(p.sym("stitching", "?buildMaps@PlaneWarper@detail@cv@@QEAA?AV?$Rect_@H@3@V?$Size_@H@3@AEBVMat@3@11AEAV63@2@Z"), obj, C.addressof(roi),
               C.addressof(sz), C.addressof(mk), C.addressof(mr), C.addressof(mt), C.addressof(xmap), C.addressof(ymap))
    else:
        p.call(p.sym("stitching", "?buildMaps@?$RotationWarperBase@U%s@detail@cv@@@detail@cv@@UEAA?AV?$Rect_@H@3@V?$Size_@H@3@AEBVMat@3@1AEAV63@2@Z" % PROJECTOR[cls]),
               obj, C.addressof(roi), C.addressof(sz), C.addressof(mk), C.addressof(mr), C.addressof(xmap), C.addressof(ymap))
    calls = p.L.pe_allow_libm(0) - before
    return {"roi": np.array(list(roi), np.int32), "xmap": mat_array(xmap), "ymap": mat_array(ymap), "libm_calls": np.array(calls, np.int64)}


def job_probe():
    p = Process()
    print(p.binding, file=sys.stderr)
    return {"ok": np.array(1)}


# ---- jobs (each runs in its own process) ---------------------------------------------------------------------------------------

def job_normalize_32f(weight, src):
    """cv::detail::normalizeUsingWeightMap(const Mat& weight, Mat& src), opencv_stitching2411.dll export at 0x1800087d0"""
    img = Image("stitching")
    w = np.ascontiguousarray(weight, np.int16 if weight.dtype == np.int16 else np.float32)      # (CV_16S weights: the `(v << 8) / (w + 1)` branch)
    s = np.ascontiguousarray(src, np.int16).copy()
    mw, ms = mat(w), mat(s)
    img.call(0x1800087D0, C.addressof(mw), C.addressof(ms))
    return {"out": s}


# cv::pyrDown / cv::pyrUp pick a function by depth (the table of `lea`s at 0x1801340d1.. / 0x180134537..): PyrFunc(const Mat& src, Mat& dst,
# int borderType) -- internal, import-free while the row buffers fit cv::AutoBuffer's 4 KB of stack (destination rows of <= 64 pixels x 3)
PYR_DOWN = {np.dtype(np.uint8): 0x18012CFF0, np.dtype(np.int16): 0x18012DB30, np.dtype(np.uint16): 0x18012E640, np.dtype(np.float32): 0x18012F150}
PYR_UP = {np.dtype(np.uint8): 0x180131160, np.dtype(np.int16): 0x180131810, np.dtype(np.uint16): 0x180131ED0, np.dtype(np.float32): 0x180132580}
BORDER_REFLECT_101 = 4


def job_pyr_down(src):
    """pyrDown_<CastOp, VecOp>(src, dst, BORDER_DEFAULT) of opencv_imgproc2411.dll; dst = ((h + 1) / 2, (w + 1) / 2) as cv::pyrDown sizes it"""
    img = Image("imgproc")
    s = np.ascontiguousarray(src)
    shape = ((s.shape[0] + 1) // 2, (s.shape[1] + 1) // 2) + s.shape[2:]
    d = np.zeros(shape, s.dtype)
    ms, md = mat(s), mat(d)
    img.call(PYR_DOWN[s.dtype], C.addressof(ms), C.addressof(md), BORDER_REFLECT_101)
    return {"out": d}


def job_pyr_up(src, dsize):
    """pyrUp_<CastOp, VecOp>(src, dst, BORDER_DEFAULT); dsize = (w, h) of the destination (|dsize - 2 ssize| <= 1)"""
    img = Image("imgproc")
    s = np.ascontiguousarray(src)
    d = np.zeros((int(dsize[1]), int(dsize[0])) + s.shape[2:], s.dtype)
    ms, md = mat(s), mat(d)
    img.call(PYR_UP[s.dtype], C.addressof(ms), C.addressof(md), BORDER_REFLECT_101)
    return {"out": d}


INIT_INTER_TAB_2D = 0x1800F7630        # static const void* initInterTab2D(int method, bool fixpt): called by cv::remap at 0x180106ee7
REMAP_BILINEAR_8U = 0x1800FE9D0        # remapBilinear<FixedPtCast<int, uchar, 15>, RemapVec_8u, short>(src, dst, xy, fxy, wtab, borderType, borderValue)
INIT_BILINEAR_TAB_IC4 = 0x1801A6AA0    # the DLL's own dynamic initializer of `BilinearTab_iC4 = alignPtr(buffer, 16)` (three instructions; the loader
                                       # would have run it -- no DllMain runs here, so the job calls the one the path needs)
INTER_LINEAR = 1


def fixed_maps(xmap, ymap):
    """float maps -> (CV_16SC2, CV_16UC1) as the loop inside cv::remap converts them (read off the binary, oracle/REFERENCE_BINARY_AUDIT.md:
    x * 32 rounded to nearest even, & 31 into the table index, >> 5 saturated to short) -- the part of remap() this runner cannot call
    (it sits between two allocations)"""
    ix = np.rint(xmap.astype(np.float32) * np.float32(32)).astype(np.int64)
    iy = np.rint(ymap.astype(np.float32) * np.float32(32)).astype(np.int64)
    ix = np.where(np.isfinite(xmap) & (np.abs(ix) < 2 ** 31), ix, -2 ** 31)           # cvtps2dq's "integer indefinite"
    iy = np.where(np.isfinite(ymap) & (np.abs(iy) < 2 ** 31), iy, -2 ** 31)
    xy = np.stack([np.clip(ix >> 5, -32768, 32767), np.clip(iy >> 5, -32768, 32767)], axis=-1).astype(np.int16)
    fxy = ((iy & 31) * 32 + (ix & 31)).astype(np.uint16)
    return np.ascontiguousarray(xy), np.ascontiguousarray(fxy)


def job_remap_bilinear(src, xy, fxy, border):
    """the 8-bit fixed-point bilinear remap of opencv_imgproc2411.dll on CV_16SC2 + CV_16UC1 maps, with the DLL's own weight table"""
    img = Image("imgproc")
    s = np.ascontiguousarray(src, np.uint8)
    m1, m2 = np.ascontiguousarray(xy, np.int16), np.ascontiguousarray(fxy, np.uint16)
    d = np.zeros(m2.shape + s.shape[2:], np.uint8)
    img.call(INIT_BILINEAR_TAB_IC4)
    tab = img.call(INIT_INTER_TAB_2D, INTER_LINEAR, 1)
    scalar = (C.c_double * 4)(0, 0, 0, 0)
    ms, md, mxy, mf = mat(s), mat(d), mat(m1), mat(m2)
    img.call(REMAP_BILINEAR_8U, C.addressof(ms), C.addressof(md), C.addressof(mxy), C.addressof(mf), tab, int(border), C.addressof(scalar))
    return {"out": d}


CVT_SCALE_8U8U = 0x18004E010           # cvtScale_<uchar, uchar, float>(src, sstep, dst, dstep, Size&, float scale, float shift)   (opencv_core2411.dll)
CVT_SCALE_8U32F = 0x180052190          # cvtScale_<uchar, float, float>(...)


def _f32bits(v):
    return int(np.float32(v).view(np.uint32))


def job_cvt_scale(src, scale, shift, to_float):
    """Mat::convertTo(dst, type, scale, shift) for 8-bit sources: what `image *= gain` and `weight.convertTo(CV_32F, 1./255.)` run"""
    img = Image("core")
    s = np.ascontiguousarray(src, np.uint8)
    s2 = s.reshape(s.shape[0], -1)
    d = np.zeros(s2.shape, np.float32 if int(to_float) else np.uint8)
    size = (C.c_int * 2)(s2.shape[1], s2.shape[0])                      # cv::Size {width, height}; the callee counts height down
    img.call(CVT_SCALE_8U32F if int(to_float) else CVT_SCALE_8U8U, s2.ctypes.data, s2.strides[0], d.ctypes.data, d.strides[0],
             C.addressof(size), _f32bits(float(scale)), _f32bits(float(shift)))
    return {"out": d.reshape(s.shape)}


DIST_3X3 = 0x1800518C0                 # icvDistanceTransform_3x3_C1R(src, srcstep, temp, tempstep, dist, diststep, CvSize, metrics): cvDistTransform's
DIST_MASK = 0x180051680                # function pointer for a 3 x 3 mask without labels; icvGetDistanceTransformMask(type, metrics)


def job_distance_l1(mask):
    """distanceTransform(mask, CV_DIST_L1, 3) as cvDistTransform runs it: the metrics from the DLL's own table (type 31 = L1 + 3 x 3 * 10),
    a (h + 2) x (w + 2) CV_32S work buffer, the 3 x 3 chamfer kernel"""
    img = Image("imgproc")
    m = np.ascontiguousarray(mask, np.uint8)
    h, w = m.shape
    metrics = (C.c_float * 4)()
    assert img.call(DIST_MASK, 31, C.addressof(metrics)) & 0xFFFFFFFF == 0
    temp = np.zeros((h + 2, w + 2), np.int32)
    dist = np.zeros((h, w), np.float32)
    img.call(DIST_3X3, m.ctypes.data, m.strides[0], temp.ctypes.data, temp.strides[0], dist.ctypes.data, dist.strides[0], w | (h << 32),
             C.addressof(metrics))
    return {"out": dist, "metrics": np.array(list(metrics)[:2], np.float32)}


NV12_TO_BGR = 0x180020180              # YUV420sp2RGB888Invoker<0, 0>::operator(): bIdx 0 (BGR), uIdx 0 (NV12) -- the first of the four instantiations
                                       # cvtColor's switch(dcn * 100 + bIdx * 10 + uIdx) picks from (0x180020540 <0,1>, 0x180020900 <2,0>, 0x180020cc0 <2,1>)


class _Nv12Invoker(C.Structure):       # YUV420sp2RGB888Invoker<bIdx, uIdx> : ParallelLoopBody  (vptr unused by operator())
    _fields_ = [("vptr", C.c_void_p), ("dst", C.c_void_p), ("my1", C.c_void_p), ("muv", C.c_void_p), ("width", C.c_int), ("stride", C.c_int)]


def job_nv12(yuv, fn):
    """one of the YUV420sp2RGB888Invoker<bIdx, uIdx>::operator()(const Range&) instantiations (fn = its address) over all row pairs"""
    img = Image("imgproc")
    y = np.ascontiguousarray(yuv, np.uint8)
    h, w = y.shape[0] * 2 // 3, y.shape[1]
    d = np.zeros((h, w, 3), np.uint8)
    md = mat(d)
    inv = _Nv12Invoker(None, C.addressof(md), y.ctypes.data, y.ctypes.data + h * y.strides[0], w, y.strides[0])
    rng = (C.c_int * 2)(0, h // 2)
    img.call(int(fn), C.addressof(inv), C.addressof(rng))
    return {"out": d}


BORDER_INTERPOLATE = 0x1800662F0       # cv::borderInterpolate(int p, int len, int borderType), an export


def job_border_interpolate(p, length, border):
    img = Image("imgproc")
    out = np.array([C.c_int32(img.call(BORDER_INTERPOLATE, int(a), int(b), int(c)) & 0xFFFFFFFF).value
                    for a, b, c in zip(p.ravel(), length.ravel(), border.ravel())], np.int32)
    return {"out": out.reshape(p.shape)}


GRAY_U8_CTOR, GRAY_U8_CALL = 0x180016C70, 0x180018520      # RGB2Gray<uchar>::RGB2Gray(int srccn, int blueIdx, const int* coeffs), ::operator()(src, dst, n)
GRAY_F32_CALL = 0x180025830                                 # RGB2Gray<float>::operator()(src, dst, n); this = {int srccn; float coeffs[3]}
GRAY_F32_COEFFS = 0x1801AB890                               # the DLL's own {0.299f, 0.587f, 0.114f}; the constructor swaps [0] and [2] for BGR input


def job_gray(src):
    """cvtColor(BGR2GRAY) of a CV_8UC3 or CV_32FC3 image: the functors cvtColor's loop calls per row"""
    img = Image("imgproc")
    s = np.ascontiguousarray(src)
    h, w = s.shape[:2]
    d = np.zeros((h, w), s.dtype)
    if s.dtype == np.uint8:
        this = (C.c_int * 769)()
        img.call(GRAY_U8_CTOR, C.addressof(this), 3, 0, 0)                    # srccn 3, blueIdx 0, default coefficients
        fn = GRAY_U8_CALL
    else:
        co = (C.c_float * 3).from_address(GRAY_F32_COEFFS)
        this = (C.c_int * 4)()
        this[0] = 3
        f = C.cast(this, C.POINTER(C.c_float))
        f[1], f[2], f[3] = co[2], co[1], co[0]
        fn = GRAY_F32_CALL
    for y in range(h):
        img.call(fn, C.addressof(this), s[y].ctypes.data, d[y].ctypes.data, w)
    return {"out": d}


ROW_FILTER_32F = 0x180069F40           # SymmRowSmallFilter<float, float, SymmRowSmallVec_32f>::operator()(src, dst, width, cn)
COL_FILTER_32F = 0x180080010           # SymmColumnSmallFilter<Cast<float, float>, SymmColumnSmallVec_32f>::operator()(src rows, dst, dststep, count, width)


def _kernel_mat(k):
    k = np.ascontiguousarray(k, np.float32).reshape(1, -1)
    return k, mat(k)


def _row_filter(k, symmetry):
    """the object as BaseRowFilter lays it out: ksize +8, anchor +0xc, Mat kernel +0x10, vecOp {Mat kernel; int symmetryType} +0x70, symmetryType +0xd8"""
    buf = (C.c_uint8 * 0x100)()
    base = C.addressof(buf)
    kk, km = _kernel_mat(k)
    C.c_int.from_address(base + 0x8).value = kk.size
    C.c_int.from_address(base + 0xC).value = kk.size // 2
    C.memmove(base + 0x10, C.addressof(km), C.sizeof(Mat))
    C.memmove(base + 0x70, C.addressof(km), C.sizeof(Mat))
    C.c_int.from_address(base + 0xD0).value = symmetry
    C.c_int.from_address(base + 0xD8).value = symmetry
    return buf, kk


def _col_filter(k, symmetry, delta=0.0):
    """ksize +8, anchor +0xc, Mat kernel +0x10, vecOp {int symmetryType; float delta; Mat kernel} +0x78, delta +0xe0, symmetryType +0xe8"""
    buf = (C.c_uint8 * 0x100)()
    base = C.addressof(buf)
    kk, km = _kernel_mat(k)
    C.c_int.from_address(base + 0x8).value = kk.size
    C.c_int.from_address(base + 0xC).value = kk.size // 2
    C.memmove(base + 0x10, C.addressof(km), C.sizeof(Mat))
    C.c_int.from_address(base + 0x78).value = symmetry
    C.c_float.from_address(base + 0x7C).value = delta
    C.memmove(base + 0x80, C.addressof(km), C.sizeof(Mat))
    C.c_float.from_address(base + 0xE0).value = delta
    C.c_int.from_address(base + 0xE8).value = symmetry
    return buf, kk


def _sep_filter_3(img, gray, kx, sx, ky, sy):
    """a 3-tap separable filter over a CV_32FC1 image with BORDER_REFLECT_101, the way FilterEngine feeds its row and column filters:
    every row with one mirrored pixel on either side through the row filter, then windows of three rows through the column filter"""
    h, w = gray.shape
    r101 = lambda p, n: 0 if n == 1 else (-p if p < 0 else (2 * n - 2 - p if p >= n else p))      # noqa: E731
    rowf, keep1 = _row_filter(kx, sx)
    colf, keep2 = _col_filter(ky, sy)
    pitch = (w + 3) // 4 * 4 + 4                                           # FilterEngine's row buffers: every row 16-byte aligned (the SSE bodies store with movaps)
    raw = np.zeros(h * pitch + 8, np.float32)
    off = (-raw.ctypes.data // 4) % 4
    tmp = raw[off:off + h * pitch].reshape(h, pitch)[:, :w]
    assert tmp.ctypes.data % 16 == 0
    for y in range(h):
        padded = np.ascontiguousarray(np.concatenate([[gray[y, r101(-1, w)]], gray[y], [gray[y, r101(w, w)]]]).astype(np.float32))
        img.call(ROW_FILTER_32F, C.addressof(rowf), padded.ctypes.data, tmp[y].ctypes.data, w, 1)
    rows = (C.c_void_p * (h + 2))(*[tmp[r101(y, h)].ctypes.data for y in range(-1, h + 1)])
    out = np.zeros((h, w), np.float32)
    img.call(COL_FILTER_32F, C.addressof(colf), C.addressof(rows), out.ctypes.data, out.strides[0], h, w)
    return out


def job_sobel32f(gray):
    """Sobel(gray, CV_32F, 1, 0) and (0, 1), ksize 3, on a CV_32FC1 image: getDerivKernels' [-1 0 1] and [1 2 1] through the DLL's own
    row and column filter objects (kernel types as getKernelType reports them: 2 = asymmetrical, 1 = symmetrical)"""
    img = Image("imgproc")
    g = np.ascontiguousarray(gray, np.float32)
    gx = _sep_filter_3(img, g, [-1, 0, 1], 2, [1, 2, 1], 1)
    gy = _sep_filter_3(img, g, [1, 2, 1], 1, [-1, 0, 1], 2)
    return {"gx": gx, "gy": gy}


def job_smooth3(src):
    """sepFilter2D(m, m, CV_32F, [.25 .5 .25], [.25 .5 .25]): the block-gain maps' smoothing (exposure_compensate.cpp:207-220)"""
    img = Image("imgproc")
    return {"out": _sep_filter_3(img, np.ascontiguousarray(src, np.float32), [0.25, 0.5, 0.25], 1, [0.25, 0.5, 0.25], 1)}


RESIZE_LINEAR_8U = 0x1800E9360         # resizeGeneric_Invoker<HResizeLinear<uchar,int,short,2048,..>, VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>,..>>::operator()
RESIZE_LINEAR_32F = 0x1800EC910        # ... <HResizeLinear<float,float,float,1,..>, VResizeLinear<float,float,float,Cast<float,float>,VResizeLinearVec_32f>>::operator()


def _linear_coeffs(sn, dn, fixpt, vertical=False):
    """cv::resize's table of source offsets and weights for INTER_LINEAR along one axis (imgwarp.cpp of 2.4.11, restated: this part of
    resize() sits between its allocations and cannot be called) -> offsets, weights (2 per output), first / one-past-last interior output"""
    scale = 1.0 / (float(dn) / sn)
    ofs = np.zeros(dn, np.int32)
    wts = np.zeros(2 * dn, np.int16 if fixpt else np.float32)
    lo, hi = 0, dn
    for d in range(dn):
        f = np.float32((d + 0.5) * scale - 0.5)
        s_ = int(np.floor(f))
        f = np.float32(f - np.float32(s_))
        if vertical:                         # rows: the offset is stored as it is (the row loop clamps it) and the weights are kept
            ofs[d] = s_
            c = (np.float32(1) - f, f)
            for k in range(2):
                wts[2 * d + k] = np.int16(np.rint(np.float64(np.float32(c[k] * np.float32(2048))))) if fixpt else c[k]
            continue
        if s_ < 0:
            lo = d + 1
            f, s_ = np.float32(0), 0
        if s_ + 1 >= sn:
            hi = min(hi, d)
            if s_ >= sn - 1:
                f, s_ = np.float32(0), sn - 1
        ofs[d] = s_
        c = (np.float32(1) - f, f)
        for k in range(2):
            wts[2 * d + k] = np.int16(np.rint(np.float64(np.float32(c[k] * np.float32(2048))))) if fixpt else c[k]
    return ofs, wts, lo, hi


def job_resize_linear(src, dsize):
    """the row loop of cv::resize(INTER_LINEAR) for a one-channel CV_8U or CV_32F image: the DLL's invoker (horizontal pass into its row ring,
    vertical pass, row clamping) on coefficient tables built as resize() builds them"""
    img = Image("imgproc")
    s = np.ascontiguousarray(src)
    fixpt = s.dtype == np.uint8
    dw, dh = int(dsize[0]), int(dsize[1])
    d = np.zeros((dh, dw), s.dtype)
    xofs, alpha, xmin, xmax = _linear_coeffs(s.shape[1], dw, fixpt)
    yofs, beta, _, _ = _linear_coeffs(s.shape[0], dh, fixpt, vertical=True)
    ms, md = mat(s), mat(d)
    obj = (C.c_uint8 * 0x110)()
    base = C.addressof(obj)
    C.memmove(base + 0x08, C.addressof(ms), C.sizeof(Mat))
    C.memmove(base + 0x68, C.addressof(md), C.sizeof(Mat))
    # (the copied headers' size / step pointers must point into the copies)
    for off in (0x08, 0x68):
        C.c_void_p.from_address(base + off + Mat.size_p.offset).value = base + off + Mat.rows.offset
        C.c_void_p.from_address(base + off + Mat.step_p.offset).value = base + off + Mat.step_buf.offset
    for off, arr in ((0xC8, xofs), (0xD0, yofs), (0xD8, alpha), (0xE0, beta)):
        C.c_void_p.from_address(base + off).value = arr.ctypes.data
    for off, v in ((0xE8, s.shape[1]), (0xEC, s.shape[0]), (0xF0, dw), (0xF4, dh), (0xF8, 2), (0xFC, xmin), (0x100, xmax)):
        C.c_int.from_address(base + off).value = v
    rng = (C.c_int * 2)(0, dh)
    img.call(RESIZE_LINEAR_8U if fixpt else RESIZE_LINEAR_32F, base, C.addressof(rng))
    return {"out": d}


REMAP_NEAREST_8U = 0x1800DE1F0         # remapNearest<uchar>(src, dst, xy, borderType, borderValue): nn_tab[CV_8U] of cv::remap (the table at 0x180218a60;
                                       # linear_tab[CV_8U] at 0x180218a20 holds REMAP_BILINEAR_8U: the dispatch the DLL itself uses)


def nearest_maps(xmap, ymap):
    """float maps -> CV_16SC2 as cv::remap converts them for INTER_NEAREST: saturate_cast<short>(x) = round to nearest even, clamped"""
    def conv(m):
        r = np.rint(m.astype(np.float32)).astype(np.float64)
        r = np.where(np.isfinite(m), r, -2.0 ** 31)
        return np.clip(r, -32768, 32767).astype(np.int16)
    return np.ascontiguousarray(np.stack([conv(xmap), conv(ymap)], axis=-1))


def job_remap_nearest(src, xy, border):
    img = Image("imgproc")
    s = np.ascontiguousarray(src, np.uint8)
    m1 = np.ascontiguousarray(xy, np.int16)
    d = np.zeros(m1.shape[:2] + s.shape[2:], np.uint8)
    scalar = (C.c_double * 4)(0, 0, 0, 0)
    ms, md, mxy = mat(s), mat(d), mat(m1)
    img.call(REMAP_NEAREST_8U, C.addressof(ms), C.addressof(md), C.addressof(mxy), int(border), C.addressof(scalar))
    return {"out": d}


BLOCKS_GAIN_APPLY = 0x140007DD0        # cv::detail::BlocksGainCompensator::apply(int index, Point corner, Mat& image, const Mat& mask) as compiled INTO THE EXE
                                       # (exposure_compensate.cpp:225-246, /fp:precise; address from the exe's own PDB)


def job_blocks_gain_apply(image, gain_map):
    """the reference's per-frame gain loop itself: with a gain map already of the image's size apply() takes `gain_map = gain_maps_[index]`
    (no resize, no allocation) and multiplies every pixel: saturate_cast<uchar>(pixel * gain)"""
    img = Image("exe")
    im = np.ascontiguousarray(image, np.uint8).copy()
    gm = np.ascontiguousarray(gain_map, np.float32)
    assert gm.shape == im.shape[:2]
    mg, mi = mat(gm), mat(im)
    mask = np.zeros((1, 1), np.uint8)
    mm = mat(mask)
    this = (C.c_uint64 * 8)()                                           # {vptr, (allocator), vector<Mat_<float>> gain_maps_ {first, last, end}}
    this[2] = C.addressof(mg)
    this[3] = this[4] = C.addressof(mg) + C.sizeof(Mat)
    corner = (C.c_int * 2)(0, 0)
    img.call(BLOCKS_GAIN_APPLY, C.addressof(this), 0, C.addressof(corner), C.addressof(mi), C.addressof(mm))
    return {"out": im}


ADD_16S, SUB_16S = 0x180018C40, 0x180019390          # add16s / sub16s(src1, step1, src2, step2, dst, step, Size&, void*): cv::add / cv::subtract on CV_16S
USE_SSE2_FLAG = 0x180266C98                          # core's `static volatile bool USE_SSE2`, set at load time by a dynamic initialiser (cpuid): not run here


def job_arith16s(a, b, sub, sse):
    """cv::subtract / cv::add on CV_16S (the Laplacian levels and their collapse): the scalar loop (the flag as an unloaded image leaves it)
    or the SSE2 one (the flag as the loader's initialiser sets it on any x86-64 CPU)"""
    img = Image("core")
    C.c_uint8.from_address(USE_SSE2_FLAG).value = 1 if int(sse) else 0
    x, y = np.ascontiguousarray(a, np.int16), np.ascontiguousarray(b, np.int16)
    x2, y2 = x.reshape(x.shape[0], -1), y.reshape(y.shape[0], -1)
    d = np.zeros(x2.shape, np.int16)
    size = (C.c_int * 2)(x2.shape[1], x2.shape[0])
    img.call(SUB_16S if int(sub) else ADD_16S, x2.ctypes.data, x2.strides[0], y2.ctypes.data, y2.strides[0], d.ctypes.data, d.strides[0], C.addressof(size), 0)
    return {"out": d.reshape(x.shape)}


JOBS = {"probe": job_probe, "build_maps": job_build_maps, "seams": job_seams, "exposure_feed": job_exposure_feed, "confinement_probe": job_confinement_probe, "pipeline": job_pipeline, "blend": job_blend, "whole": job_whole, "arith16s": job_arith16s, "blocks_gain_apply": job_blocks_gain_apply, "remap_nearest": job_remap_nearest, "resize_linear": job_resize_linear, "sobel32f": job_sobel32f, "smooth3": job_smooth3, "gray": job_gray, "border_interpolate": job_border_interpolate, "nv12": job_nv12, "distance_l1": job_distance_l1, "cvt_scale": job_cvt_scale, "normalize_32f": job_normalize_32f, "pyr_down": job_pyr_down, "pyr_up": job_pyr_up, "remap_bilinear": job_remap_bilinear}


def run(job, **arrays):
    """-> dict of output arrays, from a fresh, self-confining process"""
    if not available():
        raise RuntimeError("reference binaries are not run here (set %s=1 where /root/reference exists)" % OPT_IN)
    with tempfile.TemporaryDirectory() as d:
        fin, fout, scratch = os.path.join(d, "in.npz"), os.path.join(d, "out.npz"), os.path.join(d, "scratch")
        os.mkdir(scratch)
        os.chmod(d, 0o755)
        np.savez(fin, **arrays)
        env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), job, fin, fout, scratch], capture_output=True, text=True, timeout=900,
                           env=env)
        if r.returncode != 0:
            raise RuntimeError("ref_run %s failed (%d): %s" % (job, r.returncode, r.stderr.strip()[-600:]))
        with np.load(fout, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}


if __name__ == "__main__":
    name, fin, fout, SCRATCH = sys.argv[1:5]
    if not available():
        sys.exit("reference binaries are not run here (set %s=1 where /root/reference exists)" % OPT_IN)
    with np.load(fin, allow_pickle=False) as z:
        args = {k: z[k] for k in z.files}
    with open(fout, "wb") as answer:                             # opened BEFORE the job confines the process
        out = JOBS[name](**args)
        assert _lib().pe_locked() == 1
        np.savez(answer, **out)

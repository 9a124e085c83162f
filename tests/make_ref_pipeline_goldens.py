"""Golden PANORAMAS from the reference itself: the reference's own blender (cv::detail::MultiBandBlender / FeatherBlender / Blender of
the shipped opencv_stitching2411.dll -- constructor, prepare, feed, blend), its whole cv::remap / resize / dilate / copyMakeBorder /
pyrDown / pyrUp / distanceTransform / Mat::convertTo and the exe's BlocksGainCompensator::apply, run in place (tests/ref_run.py: opt-in,
confined) on inputs made here.  Writes

  tests/golden/ref_pipeline.npz     whole-function vectors, blender-level vectors, the reference's panorama for every committed golden
                                    case (tests/golden/{c3_div8_multiband, c2_div8_feather, c5_div16_paste_lut, stills_c1}.npz hold the
                                    inputs), and a 3-camera spherical rig with a wrap-around camera (32F and 16S blend weights)
  tests/golden/ref_stills_full.npz  two of the reference's bundled stills at their own 1920 x 1088 (REL32/output1/img-0.jpg, img-1.jpg,
                                    decoded once with Pillow), hand-set K / R, spherical, 5-band multi-band: inputs + the reference's panorama

`MST_RUN_REFERENCE_BINARIES=1 python tests/make_ref_pipeline_goldens.py` regenerates both; tests/test_reference_code_runs.py checks a
regeneration reproduces the committed bytes; tests/test_ref_pipeline.py (CPU: the oracle) and tests/test_gpu_ref_pipeline.py (the HIP
engine through the C ABI) hold both against them bit for bit.  The float warp maps handed to the reference's remap come from the
oracle's projector (sinf / cosf of MSVCR100 are not in the tree: the one unpinnable step)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import golden_util as G         # noqa: E402
import helpers as H             # noqa: E402
import oracle_lib as O          # noqa: E402  (the warp maps and the rig's seam masks: geometry, not pixels)
import ref_run as R             # noqa: E402
from stitchingvideo_amd import rig as rigmod          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_pipeline.npz")
OUT_STILLS = os.path.join(ROOT, "tests", "golden", "ref_stills_full.npz")
STILLS = "/root/reference/stitching/OpenCV2.4.11-Stitching/Release/output1"
WRAP3 = dict(n=3, size=(240, 135), warp="spherical", blend="multiband", focal=90.0, yaws=[90.0, 180.0, -90.0])
CV_16S, CV_32F = 3, 5
# product warp name -> (the reference's warper, its A / B parameters)
MAP_WARPS = {"plane": ("plane", (0, 0)), "spherical": ("spherical", (0, 0)), "cylindrical": ("cylindrical", (0, 0)), "fisheye": ("fisheye", (0, 0)),
             "stereographic": ("stereographic", (0, 0)), "compressedPlaneA2B1": ("compressedPlane", (2, 1)),
             "compressedPlaneA1.5B1": ("compressedPlane", (1.5, 1)), "compressedPlanePortraitA2B1": ("compressedPlanePortrait", (2, 1)),
             "compressedPlanePortraitA1.5B1": ("compressedPlanePortrait", (1.5, 1)), "paniniA2B1": ("panini", (2, 1)), "paniniA1.5B1": ("panini", (1.5, 1)),
             "paniniPortraitA2B1": ("paniniPortrait", (2, 1)), "paniniPortraitA1.5B1": ("paniniPortrait", (1.5, 1)), "mercator": ("mercator", (0, 0)),
             "transverseMercator": ("transverseMercator", (0, 0)), "sphericalPortrait": ("sphericalPortrait", (0, 0)),
             "cylindricalPortrait": ("cylindricalPortrait", (0, 0)), "planePortrait": ("planePortrait", (0, 0))}


def whole(fn, **a):
    return R.run("whole", fn=np.array(fn), **a)["out"]


def reference_panorama(rig, frames, blend, num_bands, sharpness, weight_type=CV_32F):
    """one frame set through the reference's compose loop (ref_run.job_pipeline) -> dict(rect, pano, pano_mask, math_xy, warped0)"""
    a, rois = {}, []
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        roi, xm, ym = O.build_maps(H.WARP_O[rig["warp"]], rig["scale"], c["K"], c["R"], f.shape[1], f.shape[0])
        rois.append(roi)
        a["frame%d" % i], a["xmap%d" % i], a["ymap%d" % i], a["mask%d" % i] = f, xm, ym, c["mask"]
        if c.get("gain_map") is not None:
            a["gain_map%d" % i] = c["gain_map"]
        a["gain%d" % i] = np.float64(c.get("gain", 1.0))
    return R.run("pipeline", warp_rois=np.array(rois), blend=np.array(blend), num_bands=np.array(num_bands), weight_type=np.array(weight_type),
                 sharpness=np.array(sharpness, np.float32), expos=np.array(rig["expos"]), **a)


def band_clamp_is_safe(out):
    """the ONE host-libm answer on the path: prepare()'s ceil(log(max_len) / log(2.0)); its argument must be far from an integer"""
    xy = out["math_xy"]
    return all(abs(x - round(x)) > 1e-6 for (fn, (x, _)) in zip(out["math_fn"], xy) if fn == b"ceil")


def cases():
    rng = np.random.default_rng(20261005)
    g = {}
    # ---- whole exported functions (allocations, prologues, tiling, borders: everything between the call and the return) ----------------
    src = rng.integers(0, 256, (211, 333, 3)).astype(np.uint8)
    xm = rng.uniform(-40, 373, (150, 420)).astype(np.float32)
    ym = rng.uniform(-40, 251, (150, 420)).astype(np.float32)
    xm[0, :6] = [-1, 0, 332, 332.5, 333, -0.5]
    ym[0, :6] = [-1, 0, 210, 210.5, 211, -0.5]
    xm[1, :3], ym[1, :3] = [np.nan, np.inf, 1e9], [3.0, 4.0, -1e9]                              # cvtps2dq's "integer indefinite"
    g["remap_src"], g["remap_xmap"], g["remap_ymap"] = src, xm, ym
    for interp, border, name in ((1, 2, "linear_reflect"), (1, 0, "linear_constant"), (0, 0, "nearest_constant"), (1, 1, "linear_replicate")):
        g["remap_out_" + name] = whole("remap", src=src, xmap=xm, ymap=ym, interp=np.array(interp), border=np.array(border))
    m = (rng.uniform(0, 1, (97, 131)) > 0.6).astype(np.uint8) * 255
    g["dilate_src"], g["dilate_out"] = m, whole("dilate", src=m)
    g["resize8_out"] = whole("resize", src=m, dsize=np.array((517, 389)))
    f = rng.uniform(0.5, 1.5, (23, 31)).astype(np.float32)
    g["resize32_src"], g["resize32_out"] = f, whole("resize", src=f, dsize=np.array((257, 189)))
    s16 = rng.integers(-3000, 3000, (95, 203, 3)).astype(np.int16)
    g["cmb_src"] = s16
    for b in (0, 1, 2, 4):
        g["cmb_out_%d" % b] = whole("copy_make_border", src=s16, tblr=np.array((7, 22, 13, 5)), border=np.array(b))
    big = rng.integers(-32768, 32767, (61, 401, 3)).astype(np.int16)             # rows far beyond the 64 pixels the leaf runs were held to
    g["pyr_src"], g["pyr_down_out"] = big, whole("pyr_down", src=big)
    up = rng.integers(-8000, 8000, (31, 201, 3)).astype(np.int16)
    g["pyr_up_src"], g["pyr_up_out"] = up, whole("pyr_up", src=up, dsize=np.array((402, 62)))
    ff = rng.uniform(0, 1, (61, 401)).astype(np.float32)
    g["pyr32_src"], g["pyr32_down_out"] = ff, whole("pyr_down", src=ff)
    mm = np.full((120, 160), 255, np.uint8)
    mm[:5] = 0
    mm[40:60, 50:90] = 0
    g["dist_src"], g["dist_out"] = mm, whole("distance_transform", src=mm)
    seam = (rng.uniform(0, 1, (60, 80)) > 0.5).astype(np.uint8) * 255
    mw = np.full((240, 317), 255, np.uint8)
    mw[:, :9] = 0
    g["fix_seam"], g["fix_mask"] = seam, mw
    g["fix_out"] = R.run("whole", fn=np.array("mask_fixup"), seam_mask=seam, mask_warped=mw)["out"]
    # ---- the blender alone: three overlapping images at odd corners (negative ones included), ragged masks ----------------------------
    corners, sizes = [(-40, -10), (20, 5), (90, -3)], [(80, 60), (90, 50), (70, 66)]
    g["blend_corners"], g["blend_sizes"] = np.array(corners), np.array(sizes)
    pl = {}
    for i, (w, h) in enumerate(sizes):
        pl["img%d" % i] = rng.integers(0, 256, (h, w, 3)).astype(np.int16)
        k = np.full((h, w), 255, np.uint8)
        k[:3, :] = 0
        k[:, :int(rng.integers(1, 9))] = 0
        k[int(rng.integers(5, h - 5)):, w - 7:] = 0
        pl["mask%d" % i] = k
        g["blend_img%d" % i], g["blend_mask%d" % i] = pl["img%d" % i], k
    x0, y0 = min(c[0] for c in corners), min(c[1] for c in corners)                             # resultRoi(corners, sizes): util.cpp:117-131
    roi = (x0, y0, max(c[0] + s[0] for c, s in zip(corners, sizes)) - x0, max(c[1] + s[1] for c, s in zip(corners, sizes)) - y0)
    g["blend_roi"] = np.array(roi)
    for name, kind, bands, wt in (("mb3_32f", "multiband", 3, CV_32F), ("mb3_16s", "multiband", 3, CV_16S), ("mb5_32f", "multiband", 5, CV_32F),
                                  ("feather", "feather", 0, CV_32F), ("no", "no", 0, CV_32F)):
        o = R.run("blend", kind=np.array(kind), corners=np.array(corners), dst_roi=np.array(roi), num_bands=np.array(bands),
                  weight_type=np.array(wt), sharpness=np.array(0.02, np.float32), **pl)
        assert band_clamp_is_safe(o), name
        g["blend_%s_dst" % name], g["blend_%s_mask" % name], g["blend_%s_pano" % name] = o["dst"], o["dst_mask"], o["pano"]
        if kind == "multiband":
            g["blend_%s_bands" % name] = o["num_bands"]
    # ---- whole frame sets: the committed golden cases' inputs through the reference's compose loop ------------------------------------
    for name in G.CASES:
        rig, frames, z = G.load(name)
        o = reference_panorama(rig, frames, rig["blend"], int(z["num_bands"]), float(z["sharpness"]))
        assert band_clamp_is_safe(o), name
        g[name + "/rect"], g[name + "/pano"], g[name + "/pano_mask"] = o["rect"], o["pano"], o["pano_mask"]
    # ---- a 3-camera spherical rig, one camera straddling the +-180 degree seam (its ROI is the whole canvas wide), block gains ---------
    rig = rigmod.build_rig(WRAP3)
    frames = rigmod.frames_for(rig, 3)
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        g["wrap3/frame%d" % i], g["wrap3/mask%d" % i], g["wrap3/gain_map%d" % i] = f, c["mask"], c["gain_map"]
        g["wrap3/K%d" % i], g["wrap3/R%d" % i] = c["K"], c["R"]
    for wt, tag in ((CV_32F, "32f"), (CV_16S, "16s")):
        o = reference_panorama(rig, frames, "multiband", 5, 0.02, wt)
        assert band_clamp_is_safe(o)
        g["wrap3/rect"], g["wrap3/pano_" + tag], g["wrap3/pano_mask_" + tag] = o["rect"], o["pano"], o["pano_mask"]
    o = reference_panorama(rig, frames, "feather", 5, 0.02)
    g["wrap3/pano_feather"], g["wrap3/pano_mask_feather"] = o["pano"], o["pano_mask"]
    # ---- the sample's --compose_megapix resize of an input: resize(full_img, img, Size(), f, f) on CV_8UC3 (CLI:699-700) ----------------
    bgr = rng.integers(0, 256, (135, 240, 3)).astype(np.uint8)
    g["resizef_src"] = bgr
    for k, f in enumerate((0.537914354, 0.25, 0.8)):
        g["resizef_f%d" % k], g["resizef_out%d" % k] = np.float64(f), whole("resize_factor", src=bgr, f=np.array(f))
    # ---- the calibration side, whole: exposure compensators' feed, seam finders' find, the warpers' buildMaps --------------------------
    corners, sizes = [(-30, -8), (25, 4), (70, -2), (110, 10)], [(80, 60), (90, 50), (70, 66), (64, 64)]
    g["cal_corners"], g["cal_sizes"] = np.array(corners), np.array(sizes)
    pl8, plf = {}, {}
    for i, (w, h) in enumerate(sizes):
        coarse = rng.uniform(0, 255, (h // 4 + 2, w // 4 + 2, 3)).astype(np.float32)          # smooth texture + noise: seams have something to follow
        img = np.ascontiguousarray(np.kron(coarse, np.ones((4, 4, 1), np.float32))[:h, :w] + rng.uniform(0, 20, (h, w, 3)).astype(np.float32))
        k = np.full((h, w), 255, np.uint8)
        k[:4] = 0
        k[:, :3] = 0
        k[h // 2:, w - 9:] = 0
        plf["img%d" % i], pl8["img%d" % i] = img, np.clip(img, 0, 255).astype(np.uint8)
        plf["mask%d" % i] = pl8["mask%d" % i] = k
        g["cal_imgf%d" % i], g["cal_img8_%d" % i], g["cal_mask%d" % i] = img, pl8["img%d" % i], k
    g["cal_gains"] = R.run("exposure_feed", kind=np.array("gain"), corners=np.array(corners), block=np.array((32, 32)), **pl8)["gains"]
    for bw, bh in ((32, 32), (16, 24)):
        r = R.run("exposure_feed", kind=np.array("gain_blocks"), corners=np.array(corners), block=np.array((bw, bh)), **pl8)
        for i in range(len(sizes)):
            g["cal_gain_map_%dx%d_%d" % (bw, bh, i)] = r["map%d" % i]
    for kind in ("voronoi", "dp_color", "dp_colorgrad"):
        for tag, pl in (("f", plf), ("8", pl8)):
            if kind == "voronoi" and tag == "8":
                continue
            r = R.run("seams", kind=np.array(kind), corners=np.array(corners), **pl)
            for i in range(len(sizes)):
                g["cal_seam_%s_%s_%d" % (kind, tag, i)] = r["mask%d" % i]
    Kc, Rc = rigmod.intrinsics(90.0, 96, 54), rigmod.rot_y(25.0)
    g["maps_K"], g["maps_R"] = Kc, Rc
    for name in MAP_WARPS:
        base, ab = MAP_WARPS[name]
        r = R.run("build_maps", warp=np.array(base), scale=np.array(90.0, np.float32), ab=np.array(ab, np.float32), K=Kc, R=Rc, size=np.array((96, 54)))
        assert (int(r["libm_calls"]) == 0) == base.startswith("plane"), name       # the plane projectors need no math library at all
        g["maps_%s_roi" % name], g["maps_%s_x" % name], g["maps_%s_y" % name] = r["roi"], r["xmap"], r["ymap"]
    # ---- the lens-undistort pre-pass as the driver runs it (DRV:201-238): initUndistortRectifyMap(K, D, Mat(), K, size, CV_16SC2) whole, then
    # remap(frame, map1, map2, INTER_LINEAR) whole on those maps; and cvtColor(COLOR_YUV2BGR_NV12) whole --------------------------------
    fw, fh = 320, 180
    Ku = np.array([[655.6 * fw / 1280, 0, 625.06 * fw / 1280], [0, 656.4 * fh / 720, 361.96 * fh / 720], [0, 0, 1]])
    dist = np.array([-0.382, 0.195, -0.00363, 0.00237])
    r = R.run("whole", fn=np.array("init_undistort"), K=Ku, dist=dist, size=np.array((fw, fh)))
    g["und_K"], g["und_dist"], g["und_map1"], g["und_map2"] = Ku, dist, r["map1"], r["map2"]
    frame = rng.integers(0, 256, (fh, fw, 3)).astype(np.uint8)
    g["und_frame"] = frame
    g["und_out"] = R.run("whole", fn=np.array("remap_fixed"), src=frame, map1=r["map1"], map2=r["map2"])["out"]
    yuv = rng.integers(0, 256, (fh * 3 // 2, fw)).astype(np.uint8)
    g["nv12w_src"], g["nv12w_out"] = yuv, R.run("whole", fn=np.array("nv12"), src=yuv)["out"]
    return g


def wrap3_rig(z):
    """the rig of the `wrap3/*` keys, in the structure the helpers use"""
    cams = [dict(K=z["wrap3/K%d" % i], R=z["wrap3/R%d" % i], size=WRAP3["size"], mask=z["wrap3/mask%d" % i],
                 gain_map=z["wrap3/gain_map%d" % i]) for i in range(3)]
    return dict(scale=WRAP3["focal"], warp="spherical", blend="multiband", expos="gain_blocks", cameras=cams, size=WRAP3["size"]), \
        [z["wrap3/frame%d" % i] for i in range(3)]


def stills_case():
    from PIL import Image
    import make_golden as MG
    stills = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(STILLS, n)).convert("RGB"))[:, :, ::-1]) for n in ("img-0.jpg", "img-1.jpg")]
    h, w = stills[0].shape[:2]
    assert (w, h) == (1920, 1088)
    rig = rigmod.build_rig(dict(n=2, size=(w, h), warp="spherical", blend="multiband", focal=1250.0, yaws=[-15.0, 15.0]), expos="no")
    o = reference_panorama(rig, stills, "multiband", 5, 0.02)
    assert band_clamp_is_safe(o)
    return MG.pack(rig, stills, "multiband", 5, 0.02, dict(rect=o["rect"], pano=o["pano"], pano_mask=o["pano_mask"]))


if __name__ == "__main__":
    if not R.available():
        sys.exit("needs /root/reference (the shipped binaries) and %s=1" % R.OPT_IN)
    np.savez_compressed(OUT, **cases())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    np.savez_compressed(OUT_STILLS, **stills_case())
    print("wrote", OUT_STILLS, os.path.getsize(OUT_STILLS), "bytes")

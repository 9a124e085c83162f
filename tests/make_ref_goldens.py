"""Golden vectors FROM THE REFERENCE ITSELF: inputs made here, outputs computed by the reference's shipped x64 machine code run in place
(tests/ref_run.py).  Runs only in a container that has /root/reference; writes tests/golden/ref_runs.npz, which travels to the GPU
box.  `python tests/make_ref_goldens.py` regenerates the file; tests/test_reference_code_runs.py checks that a regeneration
reproduces the committed bytes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402  (only for the warp maps of the rig case: geometry, not pixels)
import ref_run as R             # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_runs.npz")


def cases():
    rng = np.random.default_rng(20261004)
    g = {}
    # ---- cv::remap, 8-bit fixed-point bilinear (the warp and the lens undistort) -------------------------------------------------
    src = rng.integers(0, 256, (61, 83, 3)).astype(np.uint8)
    xm = rng.uniform(-9, 83 + 9, (45, 70)).astype(np.float32)
    ym = rng.uniform(-9, 61 + 9, (45, 70)).astype(np.float32)
    xm[0, :6] = [-1, 0, 82, 82.5, 83, -0.5]
    ym[0, :6] = [-1, 0, 60, 60.5, 61, -0.5]
    g["remap_src"], g["remap_xmap"], g["remap_ymap"] = src, xm, ym
    xy, fxy = R.fixed_maps(xm, ym)
    for border, name in ((2, "reflect"), (1, "replicate"), (4, "reflect101")):
        g["remap_out_" + name] = R.run("remap_bilinear", src=src, xy=xy, fxy=fxy, border=np.array(border))["out"]
    # BORDER_CONSTANT (the lens undistort's mode), pixels with no tap inside included (those are filled through memcpy, one of the answered imports)
    xc = rng.uniform(-5, 83 + 5, (61, 83)).astype(np.float32)
    yc = rng.uniform(-5, 61 + 5, (61, 83)).astype(np.float32)
    g["remapc_xy"], g["remapc_fxy"] = R.fixed_maps(xc, yc)
    g["remapc_out"] = R.run("remap_bilinear", src=src, xy=g["remapc_xy"], fxy=g["remapc_fxy"], border=np.array(0))["out"]
    gray = rng.integers(0, 256, (90, 120)).astype(np.uint8)
    xg = rng.uniform(-30, 150, (70, 100)).astype(np.float32)
    yg = rng.uniform(-30, 120, (70, 100)).astype(np.float32)
    g["remapg_src"], g["remapg_xmap"], g["remapg_ymap"] = gray, xg, yg
    a, b = R.fixed_maps(xg, yg)
    g["remapg_out_reflect"] = R.run("remap_bilinear", src=gray, xy=a, fxy=b, border=np.array(2))["out"]
    # INTER_NEAREST (the mask warp, BORDER_CONSTANT) on one- and three-channel images, half-way coordinates included
    xn, yn = xm.copy(), ym.copy()
    xn[1, :4] = [-0.5, 0.5, 81.5, 1.5]
    yn[1, :4] = [0.5, -0.5, 59.5, 2.5]
    g["remapn_xmap"], g["remapn_ymap"] = xn, yn
    nxy = R.nearest_maps(xn, yn)
    g["remapn_out_constant"] = R.run("remap_nearest", src=src, xy=nxy, border=np.array(0))["out"]
    g["remapn_mask_out_constant"] = R.run("remap_nearest", src=np.full((61, 83), 255, np.uint8), xy=nxy, border=np.array(0))["out"]
    g["remapn_out_reflect"] = R.run("remap_nearest", src=src, xy=nxy, border=np.array(2))["out"]
    # a whole warp: a spherical camera of a small rig, the maps from the projector, the frame through the reference's remap with BORDER_REFLECT
    K = np.array([[150.0, 0, 80.0], [0, 150.0, 45.0], [0, 0, 1]], np.float32)
    yaw = np.deg2rad(25.0)
    Rm = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]], np.float32)
    frame = rng.integers(0, 256, (90, 160, 3)).astype(np.uint8)
    roi, wx, wy = O.build_maps(1, 150.0, K, Rm, 160, 90)                # MO_WARP_SPHERICAL
    g["warp_K"], g["warp_R"], g["warp_frame"], g["warp_roi"] = K, Rm, frame, np.array(roi)
    a, b = R.fixed_maps(wx, wy)
    g["warp_out"] = R.run("remap_bilinear", src=frame, xy=a, fxy=b, border=np.array(2))["out"]
    g["warp_mask_out"] = R.run("remap_nearest", src=np.full((90, 160), 255, np.uint8), xy=R.nearest_maps(wx, wy), border=np.array(0))["out"]
    # ---- the pyramids ------------------------------------------------------------------------------------------------------------
    for i, (h, w) in enumerate(((40, 64), (37, 51), (5, 5), (2, 2), (3, 128))):
        s = rng.integers(-3000, 3000, (h, w, 3)).astype(np.int16)
        g["pd16_src_%d" % i], g["pd16_out_%d" % i] = s, R.run("pyr_down", src=s)["out"]
    for i, (h, w) in enumerate(((20, 32), (19, 26), (4, 3), (1, 1))):
        s = rng.integers(-3000, 3000, (h, w, 3)).astype(np.int16)
        g["pu16_src_%d" % i], g["pu16_out_%d" % i] = s, R.run("pyr_up", src=s, dsize=np.array((2 * w, 2 * h)))["out"]
    for i, (h, w) in enumerate(((40, 64), (33, 77), (12, 19), (2, 2))):
        s = rng.uniform(0, 1, (h, w)).astype(np.float32)
        s[rng.uniform(size=s.shape) < 0.3] = 1.0
        g["pd32_src_%d" % i], g["pd32_out_%d" % i] = s, R.run("pyr_down", src=s)["out"]
    # ---- the blender's normalisation and the conversions -------------------------------------------------------------------------
    w = rng.uniform(0, 4, (37, 53)).astype(np.float32)
    w[:3] = 0
    v = rng.integers(-32768, 32767, (37, 53, 3)).astype(np.int16)
    g["norm_weight"], g["norm_src"], g["norm_out"] = w, v, R.run("normalize_32f", weight=w, src=v)["out"]
    w16 = rng.integers(0, 1200, (23, 31)).astype(np.int16)                # MultiBandBlender(weight_type = CV_16S): weights * 256, sums up to a few cameras
    w16[:2] = 0
    v16 = rng.integers(-32768, 32767, (23, 31, 3)).astype(np.int16)
    g["norm16_weight"], g["norm16_src"], g["norm16_out"] = w16, v16, R.run("normalize_32f", weight=w16, src=v16)["out"]
    img = rng.integers(0, 256, (31, 47, 3)).astype(np.uint8)
    g["gain_src"] = img
    gains = np.array([1.0, 0.7312345, 1.2345678, 2.5, 0.999999, 1.5], np.float32)
    g["gain_values"] = gains
    for i, gv in enumerate(gains):
        g["gain_out_%d" % i] = R.run("cvt_scale", src=img, scale=np.array(gv), shift=np.array(0.0), to_float=np.array(0))["out"]
    m = rng.integers(0, 256, (20, 33)).astype(np.uint8)
    m[:4] = 255
    g["w255_src"] = m
    g["w255_out"] = R.run("cvt_scale", src=m, scale=np.array(np.float32(1. / 255.)), shift=np.array(0.0), to_float=np.array(1))["out"]
    # ---- calibration-side and ingest primitives -----------------------------------------------------------------------------------
    for i, ((h, w), p) in enumerate((((40, 64), 0.02), ((37, 51), 0.2), ((120, 200), 0.001), ((30, 30), None), ((12, 9), 2.0))):
        m = (rng.uniform(size=(h, w)) > (p or 0)).astype(np.uint8) * 255
        if p is None:
            m[0, 0] = 0                                                  # one zero in a corner: the longest chamfer paths
        g["dist_mask_%d" % i], g["dist_out_%d" % i] = m, R.run("distance_l1", mask=m)["out"]     # (p = 2: no zero at all)
    yuv = rng.integers(0, 256, (36, 40)).astype(np.uint8)                # a 40 x 24 NV12 frame, full-range bytes (clamps on both sides)
    g["nv12_src"] = yuv
    g["nv12_out"] = R.run("nv12", yuv=yuv, fn=np.array(R.NV12_TO_BGR))["out"]
    u = rng.integers(0, 256, (13, 37, 3)).astype(np.uint8)
    f = rng.uniform(0, 255, (13, 37, 3)).astype(np.float32)
    g["gray8_src"], g["gray8_out"] = u, R.run("gray", src=u)["out"]
    g["gray32_src"], g["gray32_out"] = f, R.run("gray", src=f)["out"]
    # Sobel(CV_32F, ksize 3) of the gray image above widths of every residue mod 8, and the block-gain smoothing
    for i, (h, w) in enumerate(((9, 13), (7, 8), (5, 23), (4, 9), (12, 16), (3, 1), (8, 31))):
        im = rng.uniform(0, 255, (h, w, 3)).astype(np.float32)
        gray = R.run("gray", src=im)["out"]
        r = R.run("sobel32f", gray=gray)
        g["sobel_src_%d" % i], g["sobel_gx_%d" % i], g["sobel_gy_%d" % i] = im, r["gx"], r["gy"]
    for i, (h, w) in enumerate(((32, 52), (7, 9), (3, 3), (1, 5))):
        m = rng.uniform(0.7, 1.4, (h, w)).astype(np.float32)
        g["smooth_src_%d" % i], g["smooth_out_%d" % i] = m, R.run("smooth3", src=m)["out"]
    # cv::resize(INTER_LINEAR) of a block-gain map up to a frame's size (exposure_compensate.cpp:233) and of a mask (stitcher.cpp:292)
    for i, ((sh, sw), (dw, dh)) in enumerate((((32, 52), (300, 200)), ((7, 9), (100, 64)), ((3, 3), (40, 30)), ((1, 5), (50, 7)), ((20, 33), (33, 20)))):
        m = rng.uniform(0.7, 1.4, (sh, sw)).astype(np.float32)
        u = (rng.uniform(size=(sh, sw)) > 0.4).astype(np.uint8) * 255
        g["rsz32_src_%d" % i], g["rsz32_out_%d" % i] = m, R.run("resize_linear", src=m, dsize=np.array((dw, dh)))["out"]
        g["rsz8_src_%d" % i], g["rsz8_out_%d" % i] = u, R.run("resize_linear", src=u, dsize=np.array((dw, dh)))["out"]
    # BlocksGainCompensator::apply as the reference runs it per frame (exposure_compensate.cpp:225-246): the block-gain map up-sized by the
    # DLL's resize row loop, then the exe's OWN compiled apply() over the frame
    small = rng.uniform(0.6, 1.7, (3, 5)).astype(np.float32)
    fr = rng.integers(0, 256, (72, 136, 3)).astype(np.uint8)
    full = R.run("resize_linear", src=small, dsize=np.array((136, 72)))["out"]
    g["bgain_map"], g["bgain_frame"], g["bgain_full"] = small, fr, full
    g["bgain_out"] = R.run("blocks_gain_apply", image=fr, gain_map=full)["out"]
    return g


if __name__ == "__main__":
    if not R.available():
        sys.exit("needs /root/reference (the shipped DLLs)")
    np.savez_compressed(OUT, **cases())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")

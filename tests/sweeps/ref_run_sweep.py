#!/usr/bin/env python3
"""Randomised sweep of the reference's own machine code against the oracle (tests/ref_run.py; needs /root/reference): random sizes, maps,
border modes, value ranges.  Usage: python tests/sweeps/ref_run_sweep.py [cases] [seed]   (exits non-zero on a mismatch)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import ref_run as R             # noqa: E402


def one(rng, i):
    kind = ("remap", "pyr_down16", "pyr_up16", "pyr_down32", "normalize", "gain", "distance", "nv12", "nearest", "resize32", "resize8", "sobel", "smooth")[i % 13]
    if kind == "nearest":
        sh, sw = int(rng.integers(1, 80)), int(rng.integers(1, 100))
        cn = int(rng.choice((1, 3)))
        src = rng.integers(0, 256, (sh, sw) + ((3,) if cn == 3 else ())).astype(np.uint8)
        dh, dw = int(rng.integers(1, 60)), int(rng.integers(1, 80))
        xm = rng.uniform(-sw, 2 * sw, (dh, dw)).astype(np.float32)
        ym = rng.uniform(-sh, 2 * sh, (dh, dw)).astype(np.float32)
        xm[rng.uniform(size=xm.shape) < 0.2] = np.float32(rng.integers(-3, sw + 3)) + np.float32(0.5)        # ties: round half to even
        border = int(rng.choice((0, 1, 2, 4)))
        ok = np.array_equal(R.run("remap_nearest", src=src, xy=R.nearest_maps(xm, ym), border=np.array(border))["out"], O.remap(src, xm, ym, 0, border))
        return ok, "%s %dx%dx%d -> %dx%d border %d" % (kind, sw, sh, cn, dw, dh, border)
    if kind in ("resize32", "resize8"):
        sh, sw = int(rng.integers(1, 50)), int(rng.integers(1, 70))
        dw, dh = int(rng.integers(1, 500)), int(rng.integers(1, 300))
        if kind == "resize32":
            m = rng.uniform(0.5, 2.0, (sh, sw)).astype(np.float32)
            ok = np.array_equal(R.run("resize_linear", src=m, dsize=np.array((dw, dh)))["out"].view(np.uint32), O.resize_linear_32f(m, dw, dh).view(np.uint32))
        else:
            u = rng.integers(0, 256, (sh, sw)).astype(np.uint8)
            ok = np.array_equal(R.run("resize_linear", src=u, dsize=np.array((dw, dh)))["out"], O.resize_linear_8u(u, dw, dh))
        return ok, "%s %dx%d -> %dx%d" % (kind, sw, sh, dw, dh)
    if kind == "sobel":
        h, w = int(rng.integers(1, 30)), int(rng.integers(1, 70))
        im = rng.uniform(0, 255, (h, w, 3)).astype(np.float32)
        r = R.run("sobel32f", gray=R.run("gray", src=im)["out"])
        gx, gy = O.dp_gradients(im)
        return np.array_equal(r["gx"].view(np.uint32), gx.view(np.uint32)) and np.array_equal(r["gy"].view(np.uint32), gy.view(np.uint32)), "%s %dx%d" % (kind, w, h)
    if kind == "smooth":
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        m = rng.uniform(0.5, 2.0, (h, w)).astype(np.float32)
        e = m.copy()
        O.lib().mo_smooth_gain_map.argtypes = [C.c_void_p, C.c_int, C.c_int]
        O.lib().mo_smooth_gain_map(e.ctypes.data_as(C.c_void_p), w, h)
        return np.array_equal(R.run("smooth3", src=m)["out"].view(np.uint32), e.view(np.uint32)), "%s %dx%d" % (kind, w, h)
    if kind == "remap":
        sh, sw = int(rng.integers(1, 120)), int(rng.integers(1, 160))
        cn = int(rng.choice((1, 3)))
        src = rng.integers(0, 256, (sh, sw) + ((3,) if cn == 3 else ())).astype(np.uint8)
        dh, dw = int(rng.integers(1, 90)), int(rng.integers(1, 120))
        spread = float(rng.choice((0.1, 1.0, 3.0)))
        xm = rng.uniform(-sw * spread, sw * (1 + spread), (dh, dw)).astype(np.float32)
        ym = rng.uniform(-sh * spread, sh * (1 + spread), (dh, dw)).astype(np.float32)
        border = int(rng.choice((1, 2, 4)))
        xy, fxy = R.fixed_maps(xm, ym)
        ok = np.array_equal(R.run("remap_bilinear", src=src, xy=xy, fxy=fxy, border=np.array(border))["out"], O.remap(src, xm, ym, 1, border))
        return ok, "%s %dx%dx%d -> %dx%d border %d" % (kind, sw, sh, cn, dw, dh, border)
    if kind == "pyr_down16":
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 129))
        s = rng.integers(-32768, 32767, (h, w, 3)).astype(np.int16)
        return np.array_equal(R.run("pyr_down", src=s)["out"], O.pyr_down_16s(s)), "%s %dx%d" % (kind, w, h)
    if kind == "pyr_up16":
        h, w = int(rng.integers(1, 60)), int(rng.integers(1, 55))
        s = rng.integers(-9000, 9000, (h, w, 3)).astype(np.int16)
        return np.array_equal(R.run("pyr_up", src=s, dsize=np.array((2 * w, 2 * h)))["out"], O.pyr_up_16s(s)), "%s %dx%d" % (kind, w, h)
    if kind == "pyr_down32":
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 129))
        s = rng.uniform(0, 1, (h, w)).astype(np.float32)
        return np.array_equal(R.run("pyr_down", src=s)["out"].view(np.uint32), O.pyr_down_32f(s).view(np.uint32)), "%s %dx%d" % (kind, w, h)
    if kind == "normalize":
        h, w = int(rng.integers(1, 60)), int(rng.integers(1, 90))
        wt = rng.uniform(0, 6, (h, w)).astype(np.float32)
        v = rng.integers(-32768, 32767, (h, w, 3)).astype(np.int16)
        e = v.copy()
        O.lib().mo_normalize_using_weight_map_32f(wt.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), w, h)
        return np.array_equal(R.run("normalize_32f", weight=wt, src=v)["out"], e), "%s %dx%d" % (kind, w, h)
    if kind == "gain":
        img = rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 60)), 3)).astype(np.uint8)
        gv = np.float32(rng.uniform(0.2, 4.0))
        out = R.run("cvt_scale", src=img, scale=np.array(gv), shift=np.array(0.0), to_float=np.array(0))["out"]
        return np.array_equal(out, O.gain_apply(img.copy(), float(gv))), "%s %r" % (kind, float(gv))
    if kind == "distance":
        h, w = int(rng.integers(1, 100)), int(rng.integers(1, 140))
        m = (rng.uniform(size=(h, w)) > float(rng.choice((0.001, 0.05, 0.5)))).astype(np.uint8) * 255
        return np.array_equal(R.run("distance_l1", mask=m)["out"].view(np.uint32), O.distance_transform_l1(m).view(np.uint32)), "%s %dx%d" % (kind, w, h)
    h, w = 2 * int(rng.integers(1, 30)), 2 * int(rng.integers(1, 40))
    yuv = rng.integers(0, 256, (h * 3 // 2, w)).astype(np.uint8)
    return np.array_equal(R.run("nv12", yuv=yuv, fn=np.array(R.NV12_TO_BGR))["out"], O.cvt_nv12_bgr(yuv)), "%s %dx%d" % (kind, w, h)


def main():
    if not R.available():
        sys.exit("needs /root/reference")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(cases):
        ok, what = one(rng, i)
        print("%3d %s %s" % (i, "ok  " if ok else "FAIL", what), flush=True)
        bad += not ok
    print("%d case(s), %d mismatch(es)" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Calibration-side timings on the GPU box (DRV:842-847: the reference recalibrates every 30 s; BASELINE.md section 1: its
seam finding takes 1.9-7.3 s): mst_update per blend and for a non-separable projector, mst_seam_dp (both cost functions, at
the reference's seam scale and at full frame size), mst_exposure_feed.  Prints one line per measurement."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod


def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return ts


def fmt(ts):
    return "first %.1f ms, then %s ms" % (ts[0], " / ".join("%.1f" % t for t in ts[1:]))


for name, warp in (("C3", None), ("C2", None), ("C5", None), ("C3", "stereographic"), ("C3", "fisheye")):
    cfg = dict(rigmod.CONFIGS[name])
    if warp:
        cfg["warp"] = warp                 # a non-separable projector: its backward maps are built on the host per calibration
    rig = rigmod.build_rig(cfg)
    eng = S.PanoramaEngine(len(rig["cameras"]), warp=rig["warp"], blend=cfg["blend"], expos=rig["expos"], max_batch=8)
    print("mst_update %s %s/%s: %s" % (name, rig["warp"], cfg["blend"], fmt(timed(lambda: eng.update(rig["scale"], rig["cameras"])))), flush=True)
    eng.close()

for div, label in ((5, "seam scale (384 x 216, about the reference's 0.1 Mpx)"), (1, "full frames (1920 x 1080)")):
    rig = rigmod.build_rig(rigmod.scaled_config("C3", div), seams=False)        # all-255 warped masks: the finders' input
    frames = rigmod.frames_for(rig, 0)
    warper = S.SphericalWarper(rig["scale"])
    imgs, masks = [], []
    for c, f in zip(rig["cameras"], frames):
        _, w = warper.warp(f, c["K"], c["R"])
        imgs.append(w)
        masks.append(c["mask"].copy())
    imgs_f = [im.astype(np.float32) for im in imgs]
    for cost in ("color", "color_grad"):
        def run():
            got = [m.copy() for m in masks]
            S.DpSeamFinder(cost).find(imgs_f, rig["corners"], got)
        print("mst_seam_dp C3 %s, %s: %s" % (label, cost, fmt(timed(run))), flush=True)

    def vor():
        got = [m.copy() for m in masks]
        S.VoronoiSeamFinder().find([(m.shape[1], m.shape[0]) for m in masks], rig["corners"], got)
    print("mst_seam_voronoi C3 %s: %s" % (label, fmt(timed(vor))), flush=True)
    for comp, cname in ((S.GainCompensator(), "gain"), (S.BlocksGainCompensator(), "gain_blocks")):
        try:
            print("mst_exposure_feed C3 %s, %s: %s" % (label, cname, fmt(timed(lambda: comp.feed(rig["corners"], imgs, masks)))), flush=True)
        except S.capi.StitchError as e:         # the blocks solve is bounded: full frames make more blocks than it takes
            print("mst_exposure_feed C3 %s, %s: refused (%s)" % (label, cname, e), flush=True)

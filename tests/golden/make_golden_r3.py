#!/usr/bin/env python3
"""Generate tests/golden/round3_extras.npz: small committed inputs + oracle outputs for the round-3 restatements --
DpSeamFinder (both cost functions), cvtColor(YUV2BGR_NV12), the lens-map composition and a panorama composed through it.

Run here, in the build container:  python tests/golden/make_golden_r3.py

Like the other fixtures these pin OUR restatement against regressions and give the GPU tests a committed target; they are
not reference outputs (the reference cannot run here and ships no vectors)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
import oracle_lib as O  # noqa: E402
import stitchingvideo_amd as S  # noqa: E402
from stitchingvideo_amd import rig as rigmod  # noqa: E402


def main():
    rng = np.random.default_rng(20261003)
    d = {}
    # ---- DP seams: three overlapping 8-bit images with smooth content + noise, full masks
    sizes = [(90, 60), (80, 70), (70, 64)]
    corners = [(0, 0), (55, 8), (20, 35)]
    imgs, masks = [], []
    for i, (w, h) in enumerate(sizes):
        x, y = np.meshgrid(np.arange(w), np.arange(h))
        base = 120 + 60 * np.sin((x + corners[i][0]) / 9.0) * np.cos((y + corners[i][1]) / 7.0)
        im = np.clip(base[..., None] + np.array([0, 15, 30]) + rng.integers(-12, 13, (h, w, 3)), 0, 255).astype(np.uint8)
        imgs.append(im)
        masks.append(np.full((h, w), 255, np.uint8))
    for k, cost in enumerate(("color", "color_grad")):
        out = O.dp_seams(corners, imgs, masks, cost)
        for i in range(3):
            d["dp_%s_mask%d" % (cost, i)] = out[i]
    for i in range(3):
        d["dp_img%d" % i] = imgs[i]
    d["dp_corners"] = np.array(corners, np.int32)
    # ---- NV12
    nv = rng.integers(0, 256, (36, 32)).astype(np.uint8)          # 32 x 24 frame
    d["nv12"] = nv
    d["nv12_bgr"] = O.cvt_nv12_bgr(nv)
    # ---- lens composition + a panorama through it (C3 / 16, multi-band 3 bands)
    cfg = rigmod.scaled_config("C3", 16)
    rig = rigmod.build_rig(cfg)
    w, h = rig["size"]
    K = np.array([[w * 0.52, 0, w * 0.49], [0, w * 0.52, h * 0.51], [0, 0, 1]])
    maps = [S.init_undistort_rectify_map(K, [-0.3 + 0.03 * i, 0.12, 0.001, -0.002], (w, h)) for i in range(len(rig["cameras"]))]
    frames = rigmod.frames_for(rig, 3)
    rect, pano, mask = H.oracle_compose(rig, frames, blend="multiband", num_bands=3, lens_maps=maps)
    d.update(lens_n=len(frames), lens_scale=np.float32(rig["scale"]), lens_rect=np.array(rect), lens_pano=pano, lens_pano_mask=mask)
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        d["lens_K%d" % i] = np.asarray(c["K"], np.float32); d["lens_R%d" % i] = np.asarray(c["R"], np.float32)
        d["lens_frame%d" % i] = f; d["lens_mask%d" % i] = c["mask"]; d["lens_gain_map%d" % i] = c["gain_map"]
        d["lens_map1_%d" % i] = maps[i][0]; d["lens_map2_%d" % i] = maps[i][1]
    xs = rng.uniform(-3, w + 3, 64).astype(np.float32); ys = rng.uniform(-3, h + 3, 64).astype(np.float32)
    cx, cy = O.compose_lens_maps(xs, ys, maps[0][0], maps[0][1])
    d.update(lens_probe_x=xs, lens_probe_y=ys, lens_probe_cx=cx, lens_probe_cy=cy)
    np.savez_compressed(os.path.join(HERE, "round3_extras.npz"), **d)
    print("wrote round3_extras.npz", os.path.getsize(os.path.join(HERE, "round3_extras.npz")), "bytes")


if __name__ == "__main__":
    main()

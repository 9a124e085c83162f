#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/*.npz).

Run here, in the build container:  python tests/golden/make_golden.py

What a fixture holds: the complete inputs of one compose call (K, R, scale, frames, final masks,
gain maps, configuration) and the panorama + mask the CPU oracle produces for them.  The reference
itself cannot run here (Windows binaries, OpenCV 2.4.11 absent -- SURVEY.md 8c) and ships no
golden vectors, so these pin OUR restatement against regressions and give the GPU tests a
committed target; they are not reference outputs.

`stills_c1` uses two of the reference's bundled sample frames
(/root/reference/stitching/OpenCV2.4.11-Stitching/Release/output1/img-0.jpg, img-1.jpg, 1920x1088),
decoded once here with Pillow and reduced 4x (box filter) so the GPU box needs no JPEG decoder.
Only pixel data is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
import oracle_lib as O  # noqa: E402
from stitchingvideo_amd import rig as rigmod  # noqa: E402

STILLS = "/root/reference/stitching/OpenCV2.4.11-Stitching/Release/output1"


def pack(rig, frames, blend, num_bands, sharpness, extra=None):
    d = dict(warp=rig["warp"], blend=blend, expos=rig["expos"], scale=np.float32(rig["scale"]),
             num_bands=num_bands, sharpness=np.float32(sharpness), n=len(frames))
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        d["K%d" % i] = np.asarray(c["K"], np.float32)
        d["R%d" % i] = np.asarray(c["R"], np.float32)
        d["frame%d" % i] = f
        d["mask%d" % i] = c["mask"]
        if c.get("gain_map") is not None:
            d["gain_map%d" % i] = c["gain_map"]
        d["gain%d" % i] = np.float32(c.get("gain", 1.0))
    if extra:
        d.update(extra)
    return d


def main():
    # 1. C3 / 8: 6 cameras, spherical, block gains, 5-band multi-band
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    frames = rigmod.frames_for(rig, 0)
    rect, pano, mask = H.oracle_compose(rig, frames, blend="multiband", num_bands=5)
    np.savez_compressed(os.path.join(HERE, "c3_div8_multiband.npz"),
                        **pack(rig, frames, "multiband", 5, 0.02, dict(rect=np.array(rect), pano=pano, pano_mask=mask)))

    # 2. C2 / 8: 2 cameras, plane, block gains, feather
    rig = rigmod.build_rig(rigmod.scaled_config("C2", 8))
    frames = rigmod.frames_for(rig, 1)
    rect, pano, mask = H.oracle_compose(rig, frames, blend="feather", sharpness=0.02)
    np.savez_compressed(os.path.join(HERE, "c2_div8_feather.npz"),
                        **pack(rig, frames, "feather", 5, 0.02, dict(rect=np.array(rect), pano=pano, pano_mask=mask)))

    # 3. C5 / 16: 6 cameras, cylindrical, block gains; Blender::NO and the driver's LUT composite
    rig = rigmod.build_rig(rigmod.scaled_config("C5", 16))
    frames = rigmod.frames_for(rig, 2)
    rect, pano, mask = H.oracle_compose(rig, frames, blend="no")
    imgs = []
    for c, f in zip(rig["cameras"], frames):
        _, w = O.warp(O.WARP_CYLINDRICAL, rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        imgs.append(O.blocks_gain_apply(w, c["gain_map"]).astype(np.int16))
    lut = O.Lut(rig["corners"], rig["sizes"], [c["mask"] for c in rig["cameras"]])
    lrect, lpano = lut.compose(imgs, 0.1, 0.1, 10.0, 10.0)
    np.savez_compressed(os.path.join(HERE, "c5_div16_paste_lut.npz"),
                        **pack(rig, frames, "no", 5, 0.02, dict(rect=np.array(rect), pano=pano, pano_mask=mask,
                                                                lut_rect=np.array(lrect), lut_pano=lpano,
                                                                crop=np.array([0.1, 0.1, 10.0, 10.0], np.float32))))

    # 4. C1 plumbing: two bundled stills, hand-set K/R (no registration), spherical, 5-band multi-band, no gains
    from PIL import Image
    stills = []
    for name in ("img-0.jpg", "img-1.jpg"):
        im = Image.open(os.path.join(STILLS, name)).convert("RGB")
        im = im.resize((im.width // 4, im.height // 4), Image.BOX)
        stills.append(np.ascontiguousarray(np.asarray(im)[:, :, ::-1]))         # BGR
    h, w = stills[0].shape[:2]
    cfg = dict(n=2, size=(w, h), warp="spherical", blend="multiband", focal=1250.0 / 4, yaws=[-15.0, 15.0])
    rig = rigmod.build_rig(cfg, expos="no")
    rect, pano, mask = H.oracle_compose(rig, stills, blend="multiband", num_bands=5)
    np.savez_compressed(os.path.join(HERE, "stills_c1.npz"),
                        **pack(rig, stills, "multiband", 5, 0.02, dict(rect=np.array(rect), pano=pano, pano_mask=mask)))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()

"""The reference's OWN panoramas for the BASELINE configurations at FULL SIZE (C2 2x1080p plane + feather, C3 6x1080p spherical + 5-band
multi-band, C4 8x1080p spherical + multi-band, C5 6x4K cylindrical + multi-band): the synthetic rigs and frames of stitchingvideo_amd/rig.py
through the reference's compose loop run in place (tests/ref_run.py: job_pipeline; opt-in, confined).  A panorama of 24 - 100 MB is not a
fixture: what is committed (tests/golden/ref_full_size.json) is its SHA-256, the mask's, the canvas rectangle and a CRC-32 per panorama row,
so that a mismatch names its rows.  tests/test_ref_full_size.py holds the oracle (CPU) and the engine (GPU) against it.
    MST_RUN_REFERENCE_BINARIES=1 python tests/make_ref_full_size.py"""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import make_ref_pipeline_goldens as M          # noqa: E402
import ref_run as R                            # noqa: E402
from stitchingvideo_amd import rig as rigmod   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_full_size.json")
FRAME_INDEX = {"C2": 1, "C3": 0, "C4": 2, "C5": 2}


def digest(pano, mask, rect):
    return {"rect": [int(v) for v in rect], "pano_sha256": hashlib.sha256(np.ascontiguousarray(pano).tobytes()).hexdigest(),
            "mask_sha256": hashlib.sha256(np.ascontiguousarray(mask).tobytes()).hexdigest(),
            "row_crc32": [zlib.crc32(np.ascontiguousarray(row).tobytes()) for row in pano]}


def cases():
    out = {}
    for name in ("C2", "C3", "C4", "C5"):
        cfg = rigmod.CONFIGS[name]
        rig = rigmod.build_rig(cfg)
        frames = rigmod.frames_for(rig, FRAME_INDEX[name])
        o = M.reference_panorama(rig, frames, cfg["blend"], 5, 0.02)
        assert M.band_clamp_is_safe(o), name
        out[name] = dict(digest(o["pano"], o["pano_mask"], o["rect"]), frame_index=FRAME_INDEX[name], blend=cfg["blend"], num_bands=5, sharpness=0.02)
        print(name, out[name]["rect"], out[name]["pano_sha256"][:16], flush=True)
    return out


if __name__ == "__main__":
    if not R.available():
        sys.exit("needs /root/reference (the shipped binaries) and %s=1" % R.OPT_IN)
    with open(OUT, "w") as f:
        json.dump(cases(), f, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")

"""Load a committed golden fixture (tests/golden/*.npz) back into the rig structure the helpers use."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c3_div8_multiband", "c2_div8_feather", "c5_div16_paste_lut", "stills_c1"]


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n = int(z["n"])
    cams, frames = [], []
    for i in range(n):
        f = z["frame%d" % i]
        c = dict(K=z["K%d" % i], R=z["R%d" % i], size=(f.shape[1], f.shape[0]), mask=z["mask%d" % i],
                 gain=float(z["gain%d" % i]))
        if "gain_map%d" % i in z.files:
            c["gain_map"] = z["gain_map%d" % i]
        cams.append(c)
        frames.append(f)
    rig = dict(scale=float(z["scale"]), warp=str(z["warp"]), blend=str(z["blend"]), expos=str(z["expos"]),
               cameras=cams, size=cams[0]["size"])
    return rig, frames, z

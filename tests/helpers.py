"""Shared helpers for the parity tests: run the same inputs through the oracle and the engine."""
import numpy as np

import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod

WARP_O = dict(S.capi.WARP_NAMES)      # the oracle numbers its projectors (and A/B presets) like the product does
BLEND_O = {"no": O.BLEND_NO, "feather": O.BLEND_FEATHER, "multiband": O.BLEND_MULTI_BAND}
EXPOS_O = {"no": O.EXPOS_NO, "gain": O.EXPOS_GAIN, "gain_blocks": O.EXPOS_GAIN_BLOCKS}


def oracle_compose(rig, frames, blend=None, num_bands=5, sharpness=0.02, threads=1, weight_type=None, lens_maps=None):
    """lens_maps: per camera (map1, map2) or None -- the composed single-resampling variant (frames are then RAW)."""
    cams = []
    for i, (c, f) in enumerate(zip(rig["cameras"], frames)):
        cams.append(dict(K=c["K"], R=c["R"], frame=f, mask=c["mask"], gain_map=c.get("gain_map"),
                         gain=c.get("gain", 1.0), lens_maps=None if lens_maps is None else lens_maps[i]))
    return O.compose_panorama(WARP_O[rig["warp"]], rig["scale"], BLEND_O[blend or rig["blend"]],
                              EXPOS_O[rig["expos"]], cams, num_bands=num_bands, sharpness=sharpness,
                              weight_type=O.WEIGHT_16S if weight_type == "16s" else O.WEIGHT_32F, num_threads=threads)


def make_engine(rig, blend=None, num_bands=5, sharpness=0.02, **kw):
    eng = S.PanoramaEngine(len(rig["cameras"]), warp=rig["warp"], blend=blend or rig["blend"],
                           expos=rig["expos"], num_bands=num_bands, feather_sharpness=sharpness, **kw)
    eng.update(rig["scale"], rig["cameras"])
    return eng


def diff_report(a, b):
    a = np.asarray(a).astype(np.int64)
    b = np.asarray(b).astype(np.int64)
    d = np.abs(a - b)
    return "max|d|=%d, mismatches=%d of %d" % (d.max() if d.size else 0, int((d != 0).sum()), d.size)

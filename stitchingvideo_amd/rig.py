"""Synthetic camera rigs and calibration inputs.

These are the *producers of the engine's inputs* -- what the reference's registration stage
(GetPreStruct, R64/stitching_detailed.cpp:348-694) hands to the frame loop: K, R, warp scale,
block gain maps and the final per-camera seam masks.  Registration itself is out of scope; the
rigs are analytic (cameras on a ring, R = R_y(yaw)), the masks are all-255 warped masks cut by
Voronoi seams (PairwiseSeamFinder::run + VoronoiSeamFinder::findInPair,
R64/src/seam_finders.cpp:74-85, 108-160) and the gains are the closed-form block maps of
BASELINE.md.  Nothing here is on the per-frame path.
"""
import math

import numpy as np

from . import _capi as capi
from .engine import WARPERS, synth_frame

# BASELINE.json configs (id -> parameters)
CONFIGS = {
    "C2": dict(n=2, size=(1920, 1080), warp="plane", blend="feather", focal=1250.0, yaws=[-15.0, 15.0]),
    "C3": dict(n=6, size=(1920, 1080), warp="spherical", blend="multiband", focal=1250.0,
               yaws=[0.0, 60.0, 120.0, 180.0, -120.0, -60.0]),
    "C4": dict(n=8, size=(1920, 1080), warp="spherical", blend="multiband", focal=1800.0,
               yaws=[k * 45.0 for k in range(8)]),
    "C5": dict(n=6, size=(3840, 2160), warp="cylindrical", blend="multiband", focal=2500.0,
               yaws=[0.0, 60.0, 120.0, 180.0, -120.0, -60.0]),
}


def rot_y(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64).astype(np.float32)


def intrinsics(focal, width, height, aspect=1.0):
    """CameraParams::K() (R64/src/camera.cpp:66-72) converted to CV_32F."""
    return np.array([[focal, 0, width / 2.0], [0, focal * aspect, height / 2.0], [0, 0, 1]],
                    np.float64).astype(np.float32)


def scaled_config(name, div=1):
    """A BASELINE config with the frame size and focal length divided by `div` (parity-test sizes)."""
    c = dict(CONFIGS[name])
    w, h = c["size"]
    c["size"] = (w // div, h // div)
    c["focal"] = c["focal"] / div
    return c


def warped_full_mask(warper, size, K, R):
    """RotationWarper::warp(all-255 mask, INTER_NEAREST, BORDER_CONSTANT) (stitcher.cpp:278-280),
    evaluated on the host from the engine's host map builder."""
    roi, xmap, ymap = warper.buildMaps(size, K, R, host=True)
    with np.errstate(invalid="ignore"):
        sx = np.clip(np.rint(xmap.astype(np.float64)), -32768, 32767)
        sy = np.clip(np.rint(ymap.astype(np.float64)), -32768, 32767)
    inside = (sx >= 0) & (sx < size[0]) & (sy >= 0) & (sy < size[1])
    return roi, np.where(inside, 255, 0).astype(np.uint8)


def _l1_distance_to_nonzero(img):
    """distanceTransform(img == 0, CV_DIST_L1, 3): city-block distance to the nearest nonzero pixel."""
    from scipy import ndimage
    if not img.any():
        return np.full(img.shape, 1 << 20, np.int64)
    return ndimage.distance_transform_cdt(img == 0, metric="taxicab").astype(np.int64)


def voronoi_seams(corners, masks):
    """PairwiseSeamFinder::run + VoronoiSeamFinder::findInPair, in place on `masks`."""
    n = len(masks)
    gap = 10
    for i in range(n - 1):
        for j in range(i + 1, n):
            (x1, y1), (x2, y2) = corners[i], corners[j]
            h1, w1 = masks[i].shape
            h2, w2 = masks[j].shape
            xt, yt = max(x1, x2), max(y1, y2)
            xb, yb = min(x1 + w1, x2 + w2), min(y1 + h1, y2 + h2)
            if not (xt < xb and yt < yb):
                continue
            rw, rh = xb - xt, yb - yt

            def cut(mask, tlx, tly):
                sub = np.zeros((rh + 2 * gap, rw + 2 * gap), np.uint8)
                ys, xs = yt - tly - gap, xt - tlx - gap
                y0, y1_ = max(ys, 0), min(ys + rh + 2 * gap, mask.shape[0])
                x0, x1_ = max(xs, 0), min(xs + rw + 2 * gap, mask.shape[1])
                if y0 < y1_ and x0 < x1_:
                    sub[y0 - ys:y1_ - ys, x0 - xs:x1_ - xs] = mask[y0:y1_, x0:x1_]
                return sub

            s1, s2 = cut(masks[i], x1, y1), cut(masks[j], x2, y2)
            collision = (s1 != 0) & (s2 != 0)
            u1 = np.where(collision, 0, s1)
            u2 = np.where(collision, 0, s2)
            seam = _l1_distance_to_nonzero(u1) < _l1_distance_to_nonzero(u2)
            seam = seam[gap:gap + rh, gap:gap + rw]
            m1 = masks[i][yt - y1:yt - y1 + rh, xt - x1:xt - x1 + rw]
            m2 = masks[j][yt - y2:yt - y2 + rh, xt - x2:xt - x2 + rw]
            m2[seam] = 0
            m1[~seam] = 0


def block_gain_map(cam_index, roi_w, roi_h, block=32):
    """gain_maps_[i]: 1 + 0.1*sin(i + bx/5)*cos(by/3) on ceil(W/32) x ceil(H/32) blocks."""
    gw, gh = (roi_w + block - 1) // block, (roi_h + block - 1) // block
    bx = np.arange(gw, dtype=np.float64)[None, :]
    by = np.arange(gh, dtype=np.float64)[:, None]
    return (1.0 + 0.1 * np.sin(cam_index + bx / 5.0) * np.cos(by / 3.0)).astype(np.float32)


def build_rig(cfg, expos="gain_blocks", seams=True):
    """-> dict(scale, warp, blend, cameras=[{K, R, size, roi, mask, gain_map, gain}], corners, sizes)."""
    w, h = cfg["size"]
    focal = float(cfg["focal"])
    warper = WARPERS[cfg["warp"]](focal)
    K = intrinsics(focal, w, h)
    cams, corners, masks = [], [], []
    for i, yaw in enumerate(cfg["yaws"]):
        R = rot_y(yaw)
        roi, mask = warped_full_mask(warper, (w, h), K, R)
        cams.append(dict(K=K, R=R, size=(w, h), roi=roi))
        corners.append((roi[0], roi[1]))
        masks.append(mask)
    if seams:
        voronoi_seams(corners, masks)
    for i, c in enumerate(cams):
        c["mask"] = masks[i]
        if expos == "gain_blocks":
            c["gain_map"] = block_gain_map(i, c["roi"][2], c["roi"][3])
        elif expos == "gain":
            c["gain"] = 1.0 + 0.05 * math.sin(1.0 + i)
    return dict(scale=focal, warp=cfg["warp"], blend=cfg["blend"], expos=expos, cameras=cams,
                corners=corners, sizes=[(c["roi"][2], c["roi"][3]) for c in cams], size=(w, h))


def frames_for(rig, frame_index=0):
    w, h = rig["size"]
    return [synth_frame(i, frame_index, w, h) for i in range(len(rig["cameras"]))]


def algorithmic_bytes(rig, canvas_wh):
    """every source pixel read once + every panorama pixel written once, BGR8."""
    w, h = rig["size"]
    return 3 * (len(rig["cameras"]) * w * h + canvas_wh[0] * canvas_wh[1])

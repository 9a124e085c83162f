"""tools/stitching_detailed_mi355 -- the reference's command line on the C ABI: every COMPOSITING flag of the sample is accepted
(stitching_detailed.cpp:101-119) on the CPU (--parse_only touches no GPU), and on a GPU the panorama it writes for a
--compose_megapix run is the oracle's for the same resized inputs and scaled intrinsics (CLI:662-700)."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "stitching_detailed_mi355")


def _cli(*args):
    if not os.path.exists(CLI):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "stitchingvideo_amd", "csrc"), "cli"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    return subprocess.run([CLI] + [str(a) for a in args], capture_output=True, text=True, timeout=600)


def test_every_compositing_flag_of_the_sample_is_accepted():
    for warp in ("plane", "cylindrical", "spherical", "fisheye", "stereographic", "compressedPlaneA2B1", "compressedPlaneA1.5B1",
                 "compressedPlanePortraitA2B1", "compressedPlanePortraitA1.5B1", "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1",
                 "paniniPortraitA1.5B1", "mercator", "transverseMercator"):
        r = _cli("--synthetic", 2, "640x360", "--warp", warp, "--parse_only")
        assert r.returncode == 0 and "warp %s " % warp in r.stdout, (warp, r.stdout, r.stderr)
    for expos in ("no", "gain", "gain_blocks"):
        for blend in ("no", "feather", "multiband"):
            r = _cli("--synthetic", 6, "1920x1080", "--seam_megapix", 0.1, "--seam", "voronoi", "--compose_megapix", 0.6, "--expos_comp", expos,
                     "--blend", blend, "--blend_strength", 7, "--output", "out.ppm", "--parse_only")
            assert r.returncode == 0, (r.stdout, r.stderr)
            assert "expos_comp %s blend %s blend_strength 7 output out.ppm" % (expos, blend) in r.stdout
    # --compose_megapix: min(1, sqrt(Mpx * 1e6 / area)), sizes by cvRound, intrinsics and warp scale multiplied (CLI:664-690)
    r = _cli("--synthetic", 6, "1920x1080", "--focal", 1250, "--compose_megapix", 0.6, "--parse_only")
    s = math.sqrt(0.6e6 / (1920 * 1080))
    m = re.search(r"-> (\d+)x(\d+) compose_scale (\S+) focal (\S+) ppx (\S+) ppy (\S+)", r.stdout)
    assert (int(m.group(1)), int(m.group(2))) == (round(1920 * s), round(1080 * s))
    assert abs(float(m.group(3)) - s) < 1e-8 and abs(float(m.group(4)) - 1250 * s) < 1e-5
    assert abs(float(m.group(5)) - 960 * s) < 1e-5 and abs(float(m.group(6)) - 540 * s) < 1e-5
    r = _cli("--synthetic", 6, "1920x1080", "--compose_megapix", -1, "--parse_only")           # the default: original resolution
    assert "-> 1920x1080 compose_scale 1 " in r.stdout
    r = _cli("--synthetic", 6, "1920x1080", "--compose_megapix", 1.9, "--parse_only")          # within 10 % of 1: the inputs are not resized
    assert "-> 1920x1080 " in r.stdout
    for bad in (("--expos_comp", "bogus"), ("--blend", "bogus"), ("--warp", "bogus"), ("--bogus_flag",)):
        r = _cli("--synthetic", 2, "640x360", *bad, "--parse_only")
        assert r.returncode != 0, bad


@pytest.mark.gpu
@pytest.mark.parametrize("expos,blend", [("gain", "multiband"), ("no", "feather")])
def test_compose_megapix_panorama_equals_the_oracle(tmp_path, expos, blend):
    import helpers as H
    import oracle_lib as O
    from stitchingvideo_amd import rig as rigmod
    from stitchingvideo_amd.engine import WARPERS, synth_frame
    n, fw, fh, focal, yaws, mpx = 3, 960, 540, 600.0, [-40.0, 0.0, 40.0], 0.16
    out = str(tmp_path / "pano.ppm")
    r = _cli("--synthetic", n, "%dx%d" % (fw, fh), "--focal", focal, "--yaw", ",".join(str(y) for y in yaws), "--warp", "spherical",
             "--compose_megapix", mpx, "--expos_comp", expos, "--blend", blend, "--output", out)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    s = min(1.0, math.sqrt(mpx * 1e6 / (fw * fh)))
    w, h = int(np.rint(fw * s)), int(np.rint(fh * s))
    scale = float(np.float32(focal) * np.float32(s))
    K = np.array([[focal * s, 0, fw / 2.0 * s], [0, focal * s, fh / 2.0 * s], [0, 0, 1]], np.float64).astype(np.float32)
    warper = WARPERS["spherical"](scale)
    cams = []
    for i, yaw in enumerate(yaws):
        R = rigmod.rot_y(yaw)
        roi, mask = rigmod.warped_full_mask(warper, (w, h), K, R)
        cams.append(dict(K=K, R=R, size=(w, h), roi=roi, mask=mask, gain=float(np.float32(1.0 + 0.05 * math.sin(1.0 + i))) if expos == "gain" else 1.0))
    rig = dict(scale=scale, warp="spherical", blend=blend, expos=expos, cameras=cams, size=(w, h))
    frames = [O.resize_linear_8u_factor(synth_frame(i, 0, fw, fh), s) for i in range(n)]
    kw = {}
    m = re.search(r"number of bands: (\d+)", r.stdout)
    if blend == "multiband":
        kw["num_bands"] = int(m.group(1))
    else:
        kw["sharpness"] = float(re.search(r"sharpness: (\S+)", r.stdout).group(1))
    rect, opano, _ = H.oracle_compose(rig, frames, blend=blend, **kw)
    with open(out, "rb") as f:
        assert f.readline() == b"P6\n"
        pw, ph = [int(v) for v in f.readline().split()]
        f.readline()
        rgb = np.frombuffer(f.read(), np.uint8).reshape(ph, pw, 3)
    assert (pw, ph) == rect[2:]
    if blend == "feather":                                     # (the CLI prints the sharpness with %g: compare what it ran with, not the digits)
        bw = np.sqrt(np.float32(rect[2]) * np.float32(rect[3])) * np.float32(5.0) / np.float32(100.0)      # the sample's float arithmetic (CLI:731)
        rect, opano, _ = H.oracle_compose(rig, frames, blend=blend, sharpness=float(np.float32(1.0) / bw))
    assert np.array_equal(rgb[:, :, ::-1], opano), H.diff_report(rgb[:, :, ::-1], opano)

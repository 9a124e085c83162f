"""The oracle against the committed golden fixtures (CPU): pins the restatement against regressions."""
import numpy as np
import pytest

import golden_util as G
import helpers as H
import oracle_lib as O


@pytest.mark.parametrize("name", G.CASES)
def test_oracle_reproduces_golden(name):
    rig, frames, z = G.load(name)
    rect, pano, mask = H.oracle_compose(rig, frames, blend=rig["blend"], num_bands=int(z["num_bands"]),
                                        sharpness=float(z["sharpness"]))
    assert rect == tuple(int(v) for v in z["rect"])
    assert np.array_equal(mask, z["pano_mask"])
    assert np.array_equal(pano, z["pano"])


def test_oracle_thread_count_does_not_change_results():
    rig, frames, z = G.load("c3_div8_multiband")
    _, pano, mask = H.oracle_compose(rig, frames, threads=4)
    assert np.array_equal(pano, z["pano"]) and np.array_equal(mask, z["pano_mask"])


def test_oracle_lut_golden():
    rig, frames, z = G.load("c5_div16_paste_lut")
    imgs, corners, sizes = [], [], []
    for c, f in zip(rig["cameras"], frames):
        roi, w = O.warp(O.WARP_CYLINDRICAL, rig["scale"], c["K"], c["R"], f, O.INTER_LINEAR, O.BORDER_REFLECT)
        imgs.append(O.blocks_gain_apply(w, c["gain_map"]).astype(np.int16))
        corners.append(roi[:2]); sizes.append(roi[2:])
    lut = O.Lut(corners, sizes, [c["mask"] for c in rig["cameras"]])
    rect, pano = lut.compose(imgs, *[float(v) for v in z["crop"]])
    assert rect == tuple(int(v) for v in z["lut_rect"]) and np.array_equal(pano, z["lut_pano"])


def test_native_build_matches_goldens():
    """The -O3 -march=native build that bench.py's cpu_baseline leg times (oracle/Makefile: native) is the same function:
    every golden case and the round-3 extras through it, in a child process (the flag is read when the library loads)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MST_ORACLE_NATIVE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(here, "test_golden_oracle.py"), os.path.join(here, "test_golden_round3.py"),
                        os.path.join(here, "test_oracle_kat.py"), "-k", "not native_build", "-m", "not gpu"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

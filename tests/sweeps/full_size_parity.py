#!/usr/bin/env python3
"""One frame of each BASELINE config at full size, engine vs oracle, bit for bit (the test-suite does C3 only:
the 4K oracle run takes a while).  Usage: python tests/sweeps/full_size_parity.py [C2 C4 C5 ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import helpers as H  # noqa: E402
from stitchingvideo_amd import rig as rigmod  # noqa: E402

bad = 0
for name in (sys.argv[1:] or ["C2", "C4", "C5"]):
    cfg = rigmod.CONFIGS[name]
    rig = rigmod.build_rig(cfg)
    frames = rigmod.frames_for(rig, 1)
    eng = H.make_engine(rig)
    t0 = time.perf_counter()
    pano, mask = eng.compose(frames)
    t1 = time.perf_counter()
    rect, opano, omask = H.oracle_compose(rig, frames, threads=min(32, os.cpu_count() or 1))
    t2 = time.perf_counter()
    ok = eng.canvas() == rect and np.array_equal(pano, opano) and np.array_equal(mask, omask)
    print("%s %s canvas %s  engine (host buffers) %.1f ms, oracle %.1f s" % (name, "ok" if ok else "FAIL", rect[2:], (t1 - t0) * 1e3, t2 - t1),
          flush=True)
    bad += not ok
    eng.close()
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Diagnostic: where do the blocks of kernels A and C spend their cycles?  Needs a -DMST_STAMPS build
(tools/stamps.sh builds one into a scratch copy).  Prints, per tile class, the mean shader-clock cycles between the
phase stamps of one batch-1 launch sequence.  The stamped build is never timed as a whole."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod, _capi as capi

cfg = rigmod.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rig = rigmod.build_rig(cfg)
n = len(rig["cameras"])
eng = S.PanoramaEngine(n, warp=rig["warp"], blend=cfg["blend"], expos=rig["expos"], max_batch=batch)
eng.update(rig["scale"], rig["cameras"])
(_, _, cw, ch), _ = eng.acquire()
dev = torch.device("cuda", 0)
sets = [[torch.from_numpy(f).to(dev) for f in rigmod.frames_for(rig, s)] for s in range(batch)]
panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)]
for rep in range(3):
    eng.compose_device_batch([[t.data_ptr() for t in fs] for fs in sets], [[t.stride(0) for t in fs] for fs in sets],
                             [p.data_ptr() for p in panos], panos[0].stride(0))
    eng.synchronize()
L = capi.load()
L.mst_debug_stamps.restype = C.c_int
L.mst_debug_stamps.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
for which, name, labels in ((0, "A k_warp_pyr_lds", ["tables+staging", "barrier", "sampling", "bar+tile+bar", "L0 store pass", "pyrDown"]),
                            (1, "C k_blend_coarse", ["phase1 bands", "barrier", "collapse", "level 1"]),
                            (2, "C k_blend_fine", ["level 0"])):
    buf = np.zeros(1 << 17, np.uint64)
    rc = L.mst_debug_stamps(which, buf.ctypes.data, buf.size)
    if rc:
        raise SystemExit("mst_debug_stamps rc=%d (not a -DMST_STAMPS build?)" % rc)
    st = buf.reshape(-1, 16).astype(np.int64)
    live = st[:, 0] != 0
    st = st[live]
    last = 1 if which == 2 else (6 if which == 0 else 4)
    d = np.diff(st[:, :last + 1], axis=1)
    total = st[:, last] - st[:, 0]
    print("%s: %d blocks stamped, mean lifetime %.0f cycles (p10 %.0f, p90 %.0f)" % (name, len(st), total.mean(), np.percentile(total, 10), np.percentile(total, 90)))
    if which == 2:
        entry = st[:, 8]
        own, binary, full = (entry >> 31) & 1, (entry >> 30) & 1, st[:, 9]
        ncam = np.array([bin(int(e) & 0x3FFFFFFF).count("1") for e in entry])
        classes = [("owned, full", (own == 1) & (full == 1)), ("owned, edge", (own == 1) & (full == 0)),
                   ("0/1 masks, 1 cam", (binary == 1) & (ncam == 1)), ("0/1 masks, 2 cams", (binary == 1) & (ncam == 2)),
                   ("generic, 1 cam", (own == 0) & (binary == 0) & (ncam == 1)), ("generic, 2 cams", (own == 0) & (binary == 0) & (ncam == 2)),
                   ("generic, >=3 cams", (own == 0) & (binary == 0) & (ncam >= 3)), ("no camera", ncam == 0)]
    elif which == 1:
        entry = st[:, 8]
        pure = (entry >> 31) & 1
        ncam = np.array([bin(int(e) & 0x7FFFFFFF).count("1") for e in entry])
        classes = [("pure", pure == 1), ("1 cam, not pure", (pure == 0) & (ncam == 1)), ("2 cams", ncam == 2), (">=3 cams", ncam >= 3), ("no camera", ncam == 0)]
    else:
        wb = st[:, 8]
        classes = [("all", np.ones(len(st), bool)), ("window < 24 KB", wb < 24576), ("24-40 KB", (wb >= 24576) & (wb <= 40960)), ("> 40 KB", wb > 40960)]
    for cname, m in classes:
        if not m.any():
            continue
        print("  %-16s n=%5d  total %8.0f | " % (cname, m.sum(), total[m].mean()) + "  ".join("%s %7.0f" % (lab, d[m][:, i].mean()) for i, lab in enumerate(labels)))

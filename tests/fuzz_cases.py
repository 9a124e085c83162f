"""Randomised parity cases shared by tests/test_gpu_fuzz.py (fixed-seed slice in the driver-run suite) and
tests/sweeps/fuzz_parity.py (long sweeps by hand): random rigs (warp kind, poses, sizes, focal/scale ratio),
exposure modes, mask styles, band counts, blenders, weight types, batch sizes and lens maps -- engine output
against the CPU oracle, bit for bit."""
import os

import numpy as np
import torch  # noqa: F401  (before the engine: see stitchingvideo_amd/_capi.py)

import helpers as H
import oracle_lib as O
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod
from test_gpu_edge_cases import custom_rig


def one_case(rng, idx):
    warp = ["spherical", "cylindrical", "plane"][rng.integers(0, 3)]
    n = int(rng.integers(1, 6))
    big = int(os.environ.get("FUZZ_MAX", "330"))          # largest frame width drawn (FUZZ_MAX=1400 for multi-tile cases)
    w, h = int(rng.integers(24, big)), int(rng.integers(16, max(17, big * 3 // 4)))
    focal = float(rng.uniform(0.5, 1.6) * max(w, h))
    scale = focal * float(rng.choice([1.0, 1.0, 0.6, 1.7]))
    span = 25.0 if warp == "plane" else 70.0
    poses = [(float(rng.uniform(-span, span)), float(rng.uniform(-15, 15)), float(rng.uniform(-12, 12))) for _ in range(n)]
    expos = ["no", "gain", "gain_blocks"][rng.integers(0, 3)]
    blend = ["multiband", "multiband", "feather", "no"][rng.integers(0, 4)]
    bands = int(rng.integers(1, 7))
    seams = bool(rng.integers(0, 3))           # 1/3 of the cases: overlapping full masks
    block = int(rng.choice([32, 32, 8]))
    try:
        rig = custom_rig(warp, poses, (w, h), focal, scale=scale, expos=expos, seams=seams, blend=blend, block=block)
    except Exception as e:                     # degenerate geometry (e.g. plane warp looking away): not a parity case
        return "skip (%s)" % type(e).__name__
    if max(max(c["roi"][2], c["roi"][3]) for c in rig["cameras"]) > 6000:
        return "skip (huge roi)"
    if rng.integers(0, 4) == 0:                # soft masks
        for c in rig["cameras"]:
            m = c["mask"]
            soft = rng.integers(0, 256, size=((m.shape[0] + 7) // 8, (m.shape[1] + 7) // 8)).astype(np.uint8)
            c["mask"] = np.where(m != 0, np.kron(soft, np.ones((8, 8), np.uint8))[:m.shape[0], :m.shape[1]], 0).astype(np.uint8)
    maps = [None] * n
    if rng.integers(0, 3) == 0:                # lens maps on some cameras
        K = np.array([[0.55 * w, 0, 0.5 * w], [0, 0.55 * w, 0.5 * h], [0, 0, 1]])
        for i in range(n):
            if rng.integers(0, 2):
                maps[i] = S.init_undistort_rectify_map(K, [float(rng.uniform(-0.4, 0.2)), float(rng.uniform(-0.1, 0.2)),
                                                           float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-0.01, 0.01))], (w, h))
                rig["cameras"][i]["undistort"] = maps[i]
    count = int(rng.integers(1, 4))
    wt = "16s" if (blend == "multiband" and rng.integers(0, 3) == 0) else "32f"
    # lens maps either as the reference's pre-pass (two resamplings) or composed into the warp map (one)
    composed = any(m is not None for m in maps) and bool(rng.integers(0, 2))
    # calls in flight (mst_config.overlap_depth): every fourth case runs its two calls on the engine's own streams -- they write
    # the same panoramas, so the engine has to order them (chosen from the case index: the random stream stays what it was)
    depth = (0, 3, 0, 5, 0, 0, 4, 0)[idx % 8]
    overlap = bool(rng.integers(0, 2)) and blend == "multiband"
    eng = H.make_engine(rig, blend=blend, num_bands=bands, max_batch=count, weight_type=wt,
                        overlap_frames=overlap or depth >= 3, overlap_depth=depth, lens_composed=composed)
    _, _, cw, ch = eng.canvas()
    dev = torch.device("cuda", 0)
    host = [rigmod.frames_for(rig, int(rng.integers(0, 1000))) for _ in range(count)]
    dfr = [[torch.from_numpy(f).to(dev) for f in fr] for fr in host]
    panos = [torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(count)]
    masks = [torch.zeros((ch, cw), dtype=torch.uint8, device=dev) for _ in range(count)]
    torch.cuda.synchronize()
    for rep in range(2):
        eng.compose_device_batch([[t.data_ptr() for t in fr] for fr in dfr], [[t.stride(0) for t in fr] for fr in dfr],
                                 [p.data_ptr() for p in panos], panos[0].stride(0), [m.data_ptr() for m in masks],
                                 masks[0].stride(0))
    eng.synchronize()
    ok = True
    for f in range(count):
        if composed:
            rect, opano, omask = H.oracle_compose(rig, host[f], blend=blend, num_bands=bands, weight_type=wt, lens_maps=maps)
        else:
            und = [fr if m is None else O.remap_fixed(fr, m[0], m[1]) for fr, m in zip(host[f], maps)]
            rect, opano, omask = H.oracle_compose(rig, und, blend=blend, num_bands=bands, weight_type=wt)
        ok &= eng.canvas() == rect and np.array_equal(panos[f].cpu().numpy(), opano) and np.array_equal(masks[f].cpu().numpy(), omask)
    eng.close()
    desc = "%s n=%d %dx%d f=%.0f s=%.0f %s %s/%s b=%d seams=%d batch=%d lens=%d%s flight=%d canvas=%dx%d" % (
        warp, n, w, h, focal, scale, expos, blend, wt, bands, seams, count, sum(m is not None for m in maps), "c" if composed else "", depth, cw, ch)
    return ("ok   " if ok else "FAIL ") + desc

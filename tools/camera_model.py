#!/usr/bin/env python3
"""One-GPU measurement behind DESIGN.md's multi-GPU model of the camera-sharded mode (BASELINE C4: camera i on GPU i % G,
packed payload to the root, root blends): the device time of every stage of one frame on one MI355X --

    stage(c)   mst_stage_cameras of ONE camera (warp + gain + pyramid)          runs on the camera's owner
    pack(c)    mst_payload_pack of that camera                                    owner
    unpack(c)  mst_payload_unpack                                                 root, once per remote camera
    canvas     mst_stage_canvas (both canvas kernels)                             root

and the payload bytes per camera.  From those, for G = 1, 2, 4, 8 and a stated xGMI link rate L (bytes/s, one direction,
peer -> root on its own link):

    t_owner(G) = max over ranks of sum over its cameras (stage + pack)
    t_link(G)  = max over non-root ranks of its cameras' payload bytes / L
    t_root(G)  = root's own cameras' stages + sum over remote cameras unpack + canvas
    frames/s   = 1 / max(t_owner, t_link, t_root)          (the three overlap: two pyramid and payload sets)

    python tools/camera_model.py [--link-gbs 100]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-gbs", type=float, nargs="*", default=[153.0, 100.0, 50.0],
                    help="xGMI link rates to evaluate (GB/s per direction; 153 = the link's quoted peak)")
    ap.add_argument("--config", default="C4")
    args = ap.parse_args()
    import torch
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    dev = torch.device("cuda", 0)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    rig = rigmod.build_rig(rigmod.CONFIGS[args.config])
    n = len(rig["cameras"])
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, max_batch=2)
    eng.update(rig["scale"], rig["cameras"])
    (cx, cy, cw, ch), _ = eng.acquire()
    frames = [torch.from_numpy(f).to(dev) for f in rigmod.frames_for(rig, 0)]
    ptrs = {c: frames[c].data_ptr() for c in range(n)}
    strides = {c: frames[c].stride(0) for c in range(n)}
    pano = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
    payload = [eng.payload_bytes(c) for c in range(n)]
    bufs = [torch.empty(payload[c], dtype=torch.uint8, device=dev) for c in range(n)]

    def timed(fn, reps=50):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    eng.select_stage_set(0)
    stage = [timed(lambda c=c: eng.stage_cameras([c], {c: ptrs[c]}, {c: strides[c]}, stream=stream)) for c in range(n)]
    stage_all = timed(lambda: eng.stage_cameras(list(range(n)), ptrs, strides, stream=stream))
    pack = [timed(lambda c=c: eng.payload_pack(c, bufs[c].data_ptr(), stream=stream)) for c in range(n)]
    unpack = [timed(lambda c=c: eng.payload_unpack(c, bufs[c].data_ptr(), stream=stream)) for c in range(n)]
    canvas = timed(lambda: eng.stage_canvas(pano.data_ptr(), pano.stride(0), stream=stream))
    out = {"config": args.config, "cameras": n, "canvas": [cw, ch], "payload_bytes": payload,
           "stage_one_camera_ms": [round(t * 1e3, 4) for t in stage], "stage_all_cameras_one_launch_ms": round(stage_all * 1e3, 4),
           "pack_ms": [round(t * 1e3, 4) for t in pack], "unpack_ms": [round(t * 1e3, 4) for t in unpack],
           "canvas_ms": round(canvas * 1e3, 4), "frame_sets_per_launch": 1, "predictions": []}
    for g in (1, 2, 4, 8):
        owner = [c % g for c in range(n)]
        t_owner = max(sum(stage[c] + (pack[c] if g > 1 and r != 0 else 0.0) for c in range(n) if owner[c] == r) for r in range(g))
        t_root = sum(stage[c] for c in range(n) if owner[c] == 0) + sum(unpack[c] for c in range(n) if owner[c] != 0) + canvas
        if g == 1:
            t_root = stage_all + canvas
            t_owner = t_root
        for link in args.link_gbs:
            t_link = max([sum(payload[c] for c in range(n) if owner[c] == r) / (link * 1e9) for r in range(1, g)] or [0.0])
            t = max(t_owner, t_link, t_root)
            out["predictions"].append({"gpus": g, "link_gbs": link, "t_owner_ms": round(t_owner * 1e3, 4), "t_link_ms": round(t_link * 1e3, 4),
                                       "t_root_ms": round(t_root * 1e3, 4), "frames_per_s": round(1.0 / t, 1),
                                       "bound": "owner" if t == t_owner else ("link" if t == t_link else "root")})
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()

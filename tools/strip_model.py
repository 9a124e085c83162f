#!/usr/bin/env python3
"""One-GPU measurement behind DESIGN.md's multi-GPU model: for G = 1, 2, 4, 8 column strips of the C3 / C4 canvas, the
device time of one launch sequence (8 frame sets) of every strip's engine.  With one strip per GPU the job's rate is
8 / max_k t_k panoramas/s (plus the strip exchange, see sharding.gather_strips_to_root); the sum over k against the
G = 1 time is the work the restriction adds (halo of the multi-band dependency + partial tiles)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    dev = torch.device("cuda", 0)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)          # events and launches on one stream
    stream = tstream.cuda_stream
    batch = 8
    for name in sys.argv[1:] or ["C3", "C4"]:
        rig = rigmod.build_rig(rigmod.CONFIGS[name])
        n_cam = len(rig["cameras"])
        sets = [[torch.from_numpy(f).to(dev) for f in rigmod.frames_for(rig, s)] for s in range(3)]
        ptrs = [[t.data_ptr() for t in sets[k % 3]] for k in range(batch)]
        strides = [[t.stride(0) for t in sets[k % 3]] for k in range(batch)]
        for g in (1, 2, 4, 8):
            times, widths, payload = [], [], []
            for k in range(g):
                eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, max_batch=batch, strip=(k, g))
                eng.update(rig["scale"], rig["cameras"])
                (cx, cy, cw, ch), _ = eng.acquire()
                sx, _, sw, _ = eng.strip()
                bufs = [torch.empty((ch, max(sw, 1), 3), dtype=torch.uint8, device=dev) for _ in range(batch)]
                call = eng.make_device_batch_call(ptrs, strides, [b.data_ptr() - (sx - cx) * 3 for b in bufs], bufs[0].stride(0), stream=stream)
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / 20)
                widths.append(sw)
                if k == g // 2:            # one strip's launch sequence kernel by kernel (events between the launches)
                    acc = {}
                    for _ in range(5):
                        for nm, ms in eng.profile_stages_batch(ptrs, strides, [b.data_ptr() - (sx - cx) * 3 for b in bufs], bufs[0].stride(0), stream=stream):
                            acc[nm] = acc.get(nm, 0.0) + ms / 5
                    kernels = {nm: round(v, 4) for nm, v in acc.items()}
                eng.close()
            # ... and with four launch sequences in flight per strip engine (mst_config.overlap_depth): a strip's sequence is a small
            # grid, its launch boundaries are most of what it loses
            fl_times = []
            for k in range(g):
                eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, max_batch=batch, strip=(k, g),
                                       overlap_frames=True, overlap_depth=4)
                eng.update(rig["scale"], rig["cameras"])
                (cx, cy, cw, ch), _ = eng.acquire()
                sx, _, sw, _ = eng.strip()
                ring = [[torch.empty((ch, max(sw, 1), 3), dtype=torch.uint8, device=dev) for _ in range(batch)] for _ in range(4)]
                calls = [eng.make_device_batch_call(ptrs, strides, [b.data_ptr() - (sx - cx) * 3 for b in r], r[0].stride(0), stream=stream) for r in ring]
                for i in range(8):
                    calls[i % 4]()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(40):
                    calls[i % 4]()
                e1.record()
                torch.cuda.synchronize()
                fl_times.append(e0.elapsed_time(e1) / 40)
                eng.close()
                del ring, calls
            extra = {"in_flight_4_ms_per_launch_sequence": [round(t, 4) for t in fl_times], "in_flight_4_max_ms": round(max(fl_times), 4),
                     "in_flight_4_sum_ms": round(sum(fl_times), 4),
                     "in_flight_4_predicted_frames_per_s_one_strip_per_gpu": round(batch * 1000.0 / max(fl_times), 1)}
            print(json.dumps({**extra, "config": name, "strips": g, "frame_sets_per_launch": batch, "strip_widths": widths,
                              "ms_per_launch_sequence": [round(t, 4) for t in times], "max_ms": round(max(times), 4), "sum_ms": round(sum(times), 4),
                              "predicted_frames_per_s_one_strip_per_gpu": round(batch * 1000.0 / max(times), 1),
                              "strip_bytes_to_root_per_panorama": [w * ch * 3 for w in widths], "kernels_ms_of_strip_%d" % (g // 2): kernels}), flush=True)


if __name__ == "__main__":
    main()

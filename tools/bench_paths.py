#!/usr/bin/env python3
"""Time the single-GPU BASELINE configurations other than the headline (C2 feather, C5 multi-band, C3 / C5 hard-seam
LUT, C3 no-blend paste) through the same entry points bench.py uses: frames resident in HBM, `batch` frame sets per
launch sequence, device time from HIP events on the compose stream.  One JSON object per line.

    python tools/bench_paths.py [name ...]        names: see PATHS below (default: all)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0

# name -> (BASELINE config, blend, driver_lut)
PATHS = {
    "C2": ("C2", "feather", False),
    "C5": ("C5", "multiband", False),
    "C3-LUT": ("C3", "no", True),
    "C5-LUT": ("C5", "no", True),
    "C3-paste": ("C3", "no", False),
    "C2-LUT": ("C2", "no", True),
    "C3-feather": ("C3", "feather", False),
}


_RIGS = {}


def measure(name, batch=8, min_time=0.4, latency_leg=True, expos=None, in_flight=4):
    import torch
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    cfg_name, blend, lut = PATHS[name]
    cfg = rigmod.CONFIGS[cfg_name]
    if cfg_name not in _RIGS:
        _RIGS[cfg_name] = rigmod.build_rig(cfg)
    rig = _RIGS[cfg_name]
    n_cam = len(rig["cameras"])
    dev = torch.device("cuda", 0)
    eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=expos or rig["expos"], num_bands=5, device=0,
                           driver_lut=lut, max_batch=batch)
    t0 = time.perf_counter()
    eng.update(rig["scale"], rig["cameras"])
    t_update = time.perf_counter() - t0
    (cx, cy, cw, ch), _ = eng.acquire()
    n_sets = max(3, batch) if rig["size"][0] <= 1920 else 3        # (4K rigs: three distinct frame sets, cycled)
    dev_sets = []
    for s in range(n_sets):
        fr = rigmod.frames_for(rig, frame_index=s)
        dev_sets.append([torch.from_numpy(f).to(dev) for f in fr])
    panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)]
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def tables(first):
        sets = [dev_sets[(first + k) % n_sets] for k in range(batch)]
        return [[t.data_ptr() for t in ds] for ds in sets], [[t.stride(0) for t in ds] for ds in sets]

    calls = [eng.make_device_batch_call(*tables(first), [p.data_ptr() for p in panos], panos[0].stride(0), stream=stream)
             for first in range(n_sets)]

    def timed(fn_list, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn_list[i % len(fn_list)]()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for i in range(3):
        calls[i % len(calls)]()
    torch.cuda.synchronize()
    probe = timed(calls, 5)
    n = max(10, int(min_time * 1e3 / max(probe, 1e-3)))
    ms_seq = timed(calls, n)
    alg = rigmod.algorithmic_bytes(rig, (cw, ch))
    # per-kernel durations
    acc, order = {}, []
    for i in range(5):
        st = eng.profile_stages_batch(*tables(i % n_sets), [p.data_ptr() for p in panos], panos[0].stride(0), stream=stream)
        for kname, ms in st:
            if kname not in acc:
                acc[kname] = [0.0, 0]
                order.append(kname)
            acc[kname][0] += ms
            acc[kname][1] += 1
    kernels = [{"name": k, "launches": acc[k][1] // 5, "ms_per_frame": round(acc[k][0] / 5 / batch, 5)} for k in order]
    out = {"name": name, "workload": "%s: %dx%dx%d, %s warp, gain_blocks, blend %s%s, canvas %dx%d" % (
               cfg_name, n_cam, rig["size"][0], rig["size"][1], rig["warp"], blend, " (driver LUT + crop)" if lut else "", cw, ch),
           "frame_sets_per_launch": batch, "frames_per_s": round(1000.0 * batch / ms_seq, 2),
           "ms_per_frame": round(ms_seq / batch, 5), "algorithmic_bytes_per_frame": alg,
           "roofline_frac": round(alg * batch / (ms_seq * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "kernels": kernels,
           "mst_update_s": round(t_update, 3)}
    if in_flight > 1:
        # the same calls with `in_flight` launch sequences in flight (mst_config.overlap_depth): whole calls on engine-owned
        # streams, each with its own panorama buffers; this is the rate the top-level fields report, the serial one moves to "serial"
        eng_d = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=expos or rig["expos"], num_bands=5, device=0,
                                 driver_lut=lut, max_batch=batch, overlap_frames=True, overlap_depth=max(3, in_flight))
        eng_d.update(rig["scale"], rig["cameras"])
        eng_d.acquire()
        ring = [panos] + [[torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)] for _ in range(max(3, in_flight) - 1)]
        calls_d = [eng_d.make_device_batch_call(*tables(k % n_sets), [p.data_ptr() for p in ring[k % len(ring)]], panos[0].stride(0), stream=stream)
                   for k in range(len(ring) * n_sets)]
        for i in range(2 * len(ring)):
            calls_d[i % len(calls_d)]()
        torch.cuda.synchronize()
        ms_d = timed(calls_d, max(10, int(min_time * 1e3 / max(ms_seq, 1e-3))))
        out["serial"] = {"frames_per_s": out["frames_per_s"], "ms_per_frame": out["ms_per_frame"], "roofline_frac": out["roofline_frac"]}
        out.update({"launch_sequences_in_flight": max(3, in_flight), "frames_per_s": round(1000.0 * batch / ms_d, 2),
                    "ms_per_frame": round(ms_d / batch, 5), "roofline_frac": round(alg * batch / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
        if latency_leg and batch > 1:
            one_d = [eng_d.make_device_call([t.data_ptr() for t in dev_sets[k % n_sets]], [t.stride(0) for t in dev_sets[k % n_sets]],
                                            ring[k % len(ring)][0].data_ptr(), panos[0].stride(0), stream=stream) for k in range(len(ring) * n_sets)]
            for i in range(2 * len(ring)):
                one_d[i % len(one_d)]()
            torch.cuda.synchronize()
            ms1d = timed(one_d, max(40, int(0.3 * 1e3 / max(ms_seq / batch, 1e-3))))
            out["one_frame_set_calls_in_flight"] = {"calls_in_flight": max(3, in_flight), "frames_per_s": round(1000.0 / ms1d, 2),
                                                    "roofline_frac": round(alg / (ms1d * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        eng_d.close()
        del ring, calls_d
    if latency_leg and batch > 1:
        one = [eng.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], panos[0].data_ptr(),
                                    panos[0].stride(0), stream=stream) for ds in dev_sets]
        for i in range(5):
            one[i % len(one)]()
        torch.cuda.synchronize()
        ms1 = timed(one, max(20, int(0.2 * 1e3 / max(ms_seq / batch, 1e-3))))
        # how fast the host can queue calls (no GPU wait inside the loop), and the kernels of one call by themselves
        t0 = time.perf_counter()
        for i in range(200):
            one[i % len(one)]()
        t_enq = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        ds = dev_sets[0]
        st1 = eng.profile_stages([t.data_ptr() for t in ds], [t.stride(0) for t in ds], panos[0].data_ptr(), panos[0].stride(0), stream=stream)
        out["latency_mode"] = {"ms_per_frame": round(ms1, 5), "frames_per_s": round(1000.0 / ms1, 2),
                               "roofline_frac": round(alg / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                               "host_enqueue_ms_per_call": round(t_enq * 1e3, 5), "kernels_ms": [(k, round(m, 5)) for k, m in st1]}
    eng.close()
    del dev_sets, panos
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--min-time", type=float, default=0.4)
    ap.add_argument("--no-latency-leg", action="store_true")
    ap.add_argument("--in-flight", type=int, default=4, help="launch sequences in flight for the top-level rate (1: serial only)")
    ap.add_argument("--expos", default=None, help="override the rig's exposure compensation (no | gain | gain_blocks): ablation runs")
    args = ap.parse_args()
    for nm in (args.names or list(PATHS)):
        try:
            print(json.dumps(measure(nm, batch=args.batch, min_time=args.min_time, latency_leg=not args.no_latency_leg, expos=args.expos, in_flight=args.in_flight)), flush=True)
        except Exception as e:      # noqa: BLE001
            print(json.dumps({"name": nm, "error": "%s: %s" % (type(e).__name__, e)}), flush=True)

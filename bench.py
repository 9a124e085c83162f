#!/usr/bin/env python3
"""bench.py -- stitched panorama frames/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A step is one pass of the per-frame hot path (warp -> exposure gain -> Laplacian multi-band blend
-> 8-bit panorama) over one batch of --batch (default 8) sets of N_cam synthetic frames that are already
resident in HBM: every kernel of the path is launched once per step over all frame sets of the batch
(mst_compose_device_batch) and `value` counts panoramas, i.e. steps x batch.  --batch 1 is one frame set
per launch sequence; --overlap additionally pipelines consecutive calls over two streams.  The workload at every N is BASELINE
config C3: 6x1920x1080 streams, spherical warp, block gains, 5-band multi-band blend, canvas 7852x1020.

N > 1 shards *frames*: every rank owns the full calibration and stitches its own stream of frame
sets (independent panoramas, no data-path collective) -> "scaling": "weak".

Extra objects on the JSON line:
  roofline      HBM roofline of a step: algorithmic bytes 3*(N_cam*W*H + Wc*Hc) per frame set
                x batch over the measured device time of the step; per-kernel times come from HIP events
                recorded between launches on the compose stream (mst_compose_profile_batch).
  cpu_baseline  the CPU oracle (oracle/, a port of the reference arithmetic) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--frame-sets", type=int, default=3, help="distinct synthetic frame sets cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="pipeline consecutive calls (MST_FLAG_OVERLAP_FRAMES): +2-3 %% throughput, but kernels of two "
                         "calls then run concurrently and per-kernel durations (here and in rocprofv3) stop being comparable")
    ap.add_argument("--no-overlap", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--batch", type=int, default=8,
                    help="frame sets per step (mst_compose_device_batch); 1 = one frame set per launch sequence")
    ap.add_argument("--shard", default="frames", choices=["frames", "cameras"],
                    help="N > 1: frames = every rank stitches its own frame sets (default, no collective); "
                         "cameras = camera i on rank i %% N, pyramids sent to rank 0 over RCCL, rank 0 blends (SURVEY 8e)")
    ap.add_argument("--undistort", action="store_true",
                    help="add the driver's lens-undistort pre-pass (DRV:741) to every camera (not part of the headline)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = sized to ~15 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (0 = min(host cores, 32))")
    return ap.parse_args()


def cpu_baseline(rig, frames, blend, budget_s=15.0, max_frames=0, threads=0):
    """Time the oracle on the host cores.  Test infrastructure used as the reported CPU baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads if threads > 0 else min(cores, 32)     # the row-parallel loops stop scaling past ~32 threads
    t0 = time.perf_counter()
    H.oracle_compose(rig, frames, blend=blend, threads=cores)        # warm-up + cost probe
    one = time.perf_counter() - t0
    n = max_frames if max_frames > 0 else max(1, min(30, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        H.oracle_compose(rig, frames, blend=blend, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same workload, oracle/ (C, OpenMP over cameras and rows, %d threads)" % (n, cores)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    from stitchingvideo_amd import sharding as sharding_mod

    cfg = rigmod.CONFIGS[args.config]
    rig = rigmod.build_rig(cfg)
    blend = cfg["blend"]
    n_cam = len(rig["cameras"])
    # --overlap: call t+1's per-camera kernels overlap call t's canvas kernel (frames are resident in HBM before
    # each call, as MST_FLAG_OVERLAP_FRAMES requires); the default keeps everything on one stream
    by_camera = args.shard == "cameras" and blend == "multiband"
    batch = 1 if by_camera else max(1, min(args.batch, 8))
    eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                           overlap_frames=args.overlap and not args.no_overlap and blend == "multiband" and not by_camera,
                           max_batch=batch)
    if args.undistort:
        # the reference's fisheye intrinsics (DRV:209-216, quoted for 1280x720) scaled to the frame size
        fw, fh = rig["size"]
        import numpy as np
        Ku = np.array([[655.6 * fw / 1280.0, 0, 625.06 * fw / 1280.0], [0, 656.4 * fh / 720.0, 361.96 * fh / 720.0],
                       [0, 0, 1]])
        und = S.init_undistort_rectify_map(Ku, [-0.382, 0.195, -0.00363, 0.00237], (fw, fh))
        for c in rig["cameras"]:
            c["undistort"] = und
    eng.update(rig["scale"], rig["cameras"])
    cx, cy, cw, ch = eng.canvas()

    # frames resident in HBM: a few distinct sets, each rank its own (frame index offset by rank)
    dev = torch.device("cuda", local_rank)
    host_sets, dev_sets = [], []
    n_sets = max(args.frame_sets, batch)
    for s in range(n_sets):
        fr = rigmod.frames_for(rig, frame_index=sharding_mod.global_frame_index(rank, world, s))
        host_sets.append(fr)
        dev_sets.append([torch.from_numpy(f).to(dev) for f in fr])
    panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)]
    pano = panos[0]
    # a real (non-null) stream, made torch's current one, so torch.cuda.Event brackets the launches
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    # a step = one batch of `batch` frame sets through one launch sequence (consecutive sets, rotating start)
    def tables(first):
        sets = [dev_sets[(first + k) % n_sets] for k in range(batch)]
        return [[t.data_ptr() for t in ds] for ds in sets], [[t.stride(0) for t in ds] for ds in sets]

    if by_camera:
        # camera streams sharded over ranks: a step = every rank's per-camera kernels, one exchange of the
        # Gaussian pyramids to rank 0 (point-to-point over RCCL), the canvas kernel on rank 0
        stitcher = sharding_mod.ShardedStitcher(eng, rank, world, root=0, device=dev)

        def camera_call(ds):
            ptrs = {c: ds[c].data_ptr() for c in stitcher.mine}
            strides = {c: ds[c].stride(0) for c in stitcher.mine}
            return lambda: stitcher.compose(ptrs, strides, pano.data_ptr(), pano.stride(0), stream=stream)
        calls = [camera_call(ds) for ds in dev_sets]
    elif batch == 1:
        calls = [eng.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pano.data_ptr(),
                                      pano.stride(0), stream=stream) for ds in dev_sets]
    else:
        calls = [eng.make_device_batch_call(*tables(first), [p.data_ptr() for p in panos], pano.stride(0),
                                            stream=stream) for first in range(n_sets)]

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        calls[i % len(calls)]()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        calls[i % len(calls)]()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    from stitchingvideo_amd import sharding
    wall, dev_ms = sharding.max_over_ranks([wall, dev_ms], device=dev)     # slowest rank sets the job time

    if rank == 0 and by_camera:
        frames_done = args.steps                      # one panorama per step for the whole job
        print(json.dumps({
            "metric": "stitched panorama frames/sec, 6x1080p->360deg canvas", "value": round(frames_done / wall, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1000.0 / args.steps, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8/s16 fixed-point (f32 weights)", "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d BGR8 streams, %s warp, gain_blocks, %s blend (5 bands), canvas %dx%d"
                                   % (args.config, n_cam, rig["size"][0], rig["size"][1], rig["warp"], blend, cw, ch),
                       "frames_resident_in_hbm": True,
                       "parallelism": "cameras sharded over %d rank(s), pyramids sent to rank 0 (RCCL point-to-point)" % world},
        }), flush=True)
    elif rank == 0:
        # per-kernel durations: HIP events between launches on the compose stream, averaged over a few frames
        reps = 10
        acc = {}
        order = []
        for i in range(reps):
            st = eng.profile_stages_batch(*tables(i % n_sets), [p.data_ptr() for p in panos], pano.stride(0),
                                          stream=stream)
            for name, ms in st:
                if name not in acc:
                    acc[name] = [0.0, 0]
                    order.append(name)
                acc[name][0] += ms
                acc[name][1] += 1
        kernels = [{"name": k, "launches_per_step": acc[k][1] // reps, "avg_ms": round(acc[k][0] / acc[k][1], 5),
                    "ms_per_frame": round(acc[k][0] / reps / batch, 5)} for k in order]
        dominant = max(kernels, key=lambda k: k["ms_per_frame"]) if kernels else None

        ms_per_step = wall * 1000.0 / args.steps
        dev_ms_per_step = dev_ms / args.steps
        alg_bytes = rigmod.algorithmic_bytes(rig, (cw, ch))
        traffic = None      # HBM bytes per frame from the committed rocprofv3 PMC passes of this same workload
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[args.config]["bytes_per_frame"]
        except (OSError, KeyError, ValueError):
            pass
        # one step = one launch of every kernel over `batch` frame sets: algorithmic bytes per launch = batch x frame
        achieved = alg_bytes * batch / (dev_ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "stitched panorama frames/sec, 6x1080p->360deg canvas",
            "value": round(sharding.aggregate_throughput(world, args.steps * batch, wall), 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/s16 fixed-point (f32 weights)",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d BGR8 streams, %s warp, gain_blocks, %s blend (5 bands), canvas %dx%d"
                                   % (args.config, n_cam, rig["size"][0], rig["size"][1], rig["warp"], blend, cw, ch),
                       "frames_resident_in_hbm": True, "frame_sets_per_step": batch, "lens_undistort": bool(args.undistort),
                       "parallelism": "frames sharded over %d rank(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic * batch if traffic else None,
                         "algorithmic_bytes_per_frame": alg_bytes,
                         "algorithmic_bytes_per_launch": alg_bytes * batch,
                         "device_ms_per_frame": round(dev_ms_per_step / batch, 5),
                         "scope": "whole step (%d launches over %d frame sets)"
                                  % (sum(k["launches_per_step"] for k in kernels), batch),
                         "dominant_kernel": dominant, "kernels": kernels},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rig, host_sets[0], blend, max_frames=args.cpu_frames, threads=args.cpu_threads)
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

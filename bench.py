#!/usr/bin/env python3
"""bench.py -- stitched panorama frames/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process spawns the N ranks itself (torch.distributed.run,
127.0.0.1 rendezvous) BEFORE anything touches a GPU and forwards rank 0's JSON line; when the driver launches
the ranks (RANK / WORLD_SIZE set) it runs as one of them.

Workload (every N): BASELINE config C3 -- 6x1920x1080 streams, spherical warp, block gains, 5-band multi-band
blend, canvas 7852x1020 -- with the frames already resident in HBM.

A step is one pass of the per-frame hot path (warp -> exposure gain -> Laplacian multi-band blend -> 8-bit
panorama) over one batch of synthetic input.  The batch of a step is `reps` consecutive launch sequences of
--batch (default 8) independent frame sets each (mst_compose_device_batch: every kernel of the path launched once
over the 8 sets); `reps` is sized once, from the warm-up, so that the K timed steps last at least --min-time
(1 s): exactly K steps are timed, and a 15 ms timed region (20 steps of one launch sequence) is a thing of the
past.  `value` counts panoramas: K x reps x batch x N / wall.

N > 1, default: *frames* are sharded -- every rank owns the full calibration and stitches its own stream of frame
sets (independent panoramas, no data-path collective) -> "scaling": "weak".  The north-star split (camera i on GPU
i % N, one exchange of the per-camera pyramid payload to rank 0 over RCCL, rank 0 blends: SURVEY 8e, BASELINE C4)
is measured in the same run on the C4 rig and reported in the `camera_sharded` object of the same line (or as the
headline with --shard cameras).

Extra objects on the JSON line:
  roofline      HBM roofline of the dominant kernel and of the whole step: algorithmic bytes
                3*(N_cam*W*H + Wc*Hc) per frame set x frame sets per launch over the launch duration,
                measured live with HIP events recorded between the launches on the compose stream
                (mst_compose_profile_batch); `traffic` = HBM bytes per launch from the rocprofv3 PMC passes in
                profiles/traffic.json, reported only while that file was produced from the current kernel sources.
  latency_mode  one frame set per launch sequence (the reference's one StitchingAll() per frame), same engine.
  cpu_baseline  the CPU oracle (oracle/, a port of the reference arithmetic) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
METRIC = "stitched panorama frames/sec, 6x1080p->360deg canvas"
printed = False          # rank 0 has printed its JSON line (camera-leg watchdog)
KERNEL_SOURCES = ["stitchingvideo_amd/csrc/kernels_mb.hip", "stitchingvideo_amd/csrc/kernels.hip",
                  "stitchingvideo_amd/csrc/kernels_compose.hip", "stitchingvideo_amd/csrc/kernels_common.hpp",
                  "stitchingvideo_amd/csrc/device_types.hpp"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--min-time", type=float, default=1.0, help="the K timed steps last at least this long (seconds)")
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--frame-sets", type=int, default=3, help="distinct synthetic frame sets cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="pipeline consecutive calls (MST_FLAG_OVERLAP_FRAMES): kernels of two calls then run "
                         "concurrently and per-kernel durations (here and in rocprofv3) stop being comparable")
    ap.add_argument("--no-overlap", action="store_true", help="(kept for older command lines: same as --in-flight 1)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="launch sequences in flight in the timed region (mst_config.overlap_depth: whole calls on engine-owned "
                         "streams, each with its own panorama buffers); 1 = strictly one after the other, what the per-kernel "
                         "durations and the rocprofv3 profiles are taken with")
    ap.add_argument("--batch", type=int, default=8,
                    help="frame sets per launch sequence (mst_compose_device_batch); 1 = one frame set per sequence")
    ap.add_argument("--strip-leg", action="store_true", help="run the strip-sharded measurement at N = 1 too (one strip = the whole canvas)")
    ap.add_argument("--strip-headline", action="store_true",
                    help="N = 1: time the headline through the N > 1 code path (this rank's engine composes strip 0 of 1 = the whole canvas)")
    ap.add_argument("--shard", default="frames", choices=["frames", "cameras"],
                    help="N > 1 headline: frames = every rank stitches its own frame sets (default, no collective); "
                         "cameras = camera i on rank i %% N, payload sent to rank 0 over RCCL, rank 0 blends (SURVEY 8e)")
    ap.add_argument("--pano-row-pad", type=int, default=0, help="extra bytes per panorama row (stride experiments)")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the one-frame-set-per-sequence measurement (profiling runs)")
    ap.add_argument("--no-camera-leg", action="store_true", help="N > 1: skip the camera-sharded secondary measurement")
    ap.add_argument("--camera-leg", action="store_true", help="run the camera-sharded measurement at N = 1 too (stage split only)")
    ap.add_argument("--camera-leg-steps", type=int, default=100)
    ap.add_argument("--undistort", action="store_true",
                    help="add the driver's lens-undistort pre-pass (DRV:741) to every camera (not part of the headline)")
    ap.add_argument("--lens-composed", action="store_true",
                    help="with --undistort: compose the lens maps into the warp's map instead (MST_FLAG_LENS_COMPOSED: one "
                         "resampling, no pre-pass; the fused variant's own pixels, not the reference's)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = sized to ~15 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (0 = min(host cores, 32))")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` object (the other single-GPU BASELINE configurations: C2 feather, C5 multi-band, "
                         "C3 / C5 driver LUT), e.g. under a profiler")
    return ap.parse_args()


def spawn_ranks(n):
    """--gpus N from a plain shell: start the N ranks as child processes and hand back their exit code.  Nothing in
    this process has touched a GPU (torch.cuda.device_count() does not initialise one on this image), and nothing
    is exec'ed: the children are ordinary subprocesses."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class ClockSampler:
    """The shader clock and board power while the timed region runs, from sysfs (tools/clock_watch.py does the same around any
    command): a box may show several cards, the busiest one is the job's.  Never touches the HIP runtime."""

    def __init__(self, period=0.02):
        import glob
        self.period, self.cards = period, {}
        for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            card = hw.split("/")[4]
            src = {}
            for key, names in (("sclk_hz", ["freq1_input"]), ("power_uw", ["power1_average", "power1_input"])):
                for n in names:
                    if os.path.exists("%s/%s" % (hw, n)):
                        src.setdefault(key, "%s/%s" % (hw, n))
            busy = "/sys/class/drm/%s/device/gpu_busy_percent" % card
            if os.path.exists(busy):
                src["busy_pct"] = busy
            if "sclk_hz" in src and "busy_pct" in src:
                self.cards[card] = src
        self.samples = {c: {k: [] for k in src} for c, src in self.cards.items()}
        self.stop = threading.Event()
        self.thread = None

    def __enter__(self):
        def loop():
            while not self.stop.is_set():
                for c, src in self.cards.items():
                    for k, p in src.items():
                        try:
                            self.samples[c][k].append(int(open(p).read().strip()))
                        except (OSError, ValueError):
                            self.samples[c][k].append(-1)
                time.sleep(self.period)
        if self.cards:
            self.thread = threading.Thread(target=loop, daemon=True)
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.thread:
            self.thread.join()

    def report(self):
        def mean_busy(c):
            b = [x for x in self.samples[c]["busy_pct"] if x >= 0]
            return sum(b) / len(b) if b else -1.0
        if not self.cards:
            return None
        card = max(self.cards, key=mean_busy)
        sm = self.samples[card]
        hot = [i for i, b in enumerate(sm["busy_pct"]) if b >= 90]
        if len(hot) < 3:
            return None
        out = {"card": card, "samples_under_load": len(hot)}
        for k in ("sclk_hz", "power_uw"):
            v = sorted(sm[k][i] for i in hot if k in sm and i < len(sm[k]) and sm[k][i] >= 0)
            if v:
                out[k.replace("_hz", "_mhz").replace("_uw", "_w")] = {"median": round(v[len(v) // 2] / 1e6, 1), "min": round(v[0] / 1e6, 1), "max": round(v[-1] / 1e6, 1)}
        return out


def kernel_source_hash():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline(rig, frames, blend, budget_s=15.0, max_frames=0, threads=0):
    """Time the oracle on the host cores: all cores (capped at 32: its row-parallel loops stop scaling there) and ONE
    thread.  Test infrastructure used as the reported CPU baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # -O3 -march=native (contraction off), compiled on this host by oracle/Makefile's `native` target;
    # the portable -O2 build the tests use is the fallback when no compiler is at hand
    os.environ["MST_ORACLE_NATIVE"] = "1"
    build = "-O3 -march=native -fopenmp -ffp-contract=off, built on this host"
    try:
        import oracle_lib
        oracle_lib.lib()
    except Exception:
        os.environ["MST_ORACLE_NATIVE"] = "0"
        build = "-O2 -fopenmp -ffp-contract=off (portable build; the native target failed to compile here)"
    import helpers as H
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads if threads > 0 else min(cores, 32)

    def sample(nthreads, budget, lo, hi):
        t0 = time.perf_counter()
        H.oracle_compose(rig, frames, blend=blend, threads=nthreads)        # warm-up + cost probe
        one = time.perf_counter() - t0
        n = max_frames if max_frames > 0 else max(lo, min(hi, int(budget / max(one, 1e-3))))
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            H.oracle_compose(rig, frames, blend=blend, threads=nthreads)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        pick = lambda q: ts[min(len(ts) - 1, int(q * len(ts)))]
        # fps percentiles: p10 = slow tail, p90 = fast tail
        return {"frames": n, "median_fps": round(1.0 / pick(0.5), 4), "p10_fps": round(1.0 / pick(0.9), 4),
                "p90_fps": round(1.0 / pick(0.1), 4), "mean_fps": round(n / sum(ts), 4)}

    allc = sample(cores, budget_s, 5, 30)
    one = sample(1, budget_s, 3, 5) if cores > 1 else allc
    return {"value": allc["mean_fps"], "unit": "frames/s", "cores": cores, "kind": "port", "build": build,
            "median": allc["median_fps"], "p10": allc["p10_fps"], "p90": allc["p90_fps"],
            "threads_1": {"value": one["median_fps"], "p10": one["p10_fps"], "p90": one["p90_fps"], "frames": one["frames"]},
            "sample": "%d frames of the same workload at %d threads and %d at 1 thread, oracle/ (C, OpenMP over cameras and rows); "
                      "median / p10 / p90 of the per-frame rates" % (allc["frames"], cores, one["frames"])}


def main():
    global printed
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # MST_BENCH_SMOKE_BACKEND=gloo: a test hook for boxes with fewer GPUs than ranks -- the ranks share the visible devices and the
    # timing reductions travel over gloo on host tensors.  It exercises the N > 1 code path (strip-mode headline, replicas leg);
    # its numbers mean nothing and its line says so.  RCCL refuses two ranks of a communicator on one device.
    smoke_backend = os.environ.get("MST_BENCH_SMOKE_BACKEND", "") if world > 1 else ""
    if smoke_backend:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1 and smoke_backend:
        dist.init_process_group(backend=smoke_backend)
    elif world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    from stitchingvideo_amd import sharding

    cfg = rigmod.CONFIGS[args.config]
    rig = rigmod.build_rig(cfg)
    blend = cfg["blend"]
    n_cam = len(rig["cameras"])
    by_camera = args.shard == "cameras" and blend == "multiband"
    batch = 1 if by_camera else max(1, min(args.batch, 8))
    # N > 1: `value` keeps the meaning it has at N = 1 -- WHOLE panoramas delivered per second -- with per-GPU work fixed ("weak"): the
    # frame sets are the independent units, rank r stitches frame sets r, r + N, r + 2N ... into whole panoramas in its own HBM, no
    # data-path collective.  The modes that split ONE panorama over the GPUs are reported beside it under their own keys, each with
    # its gather to one consumer inside the timed region: `camera_sharded` (the north star's C4 split: camera i on GPU i mod N, one
    # grouped RCCL send/recv of the pyramid payloads to rank 0) and `strip_sharded` (the canvas cut into N column strips).
    # --strip-headline puts the strip mode's strips-left-in-place rate into `value` instead (a one-GPU test hook, labelled "strong").
    strip_mode = args.strip_headline and not by_camera and blend == "multiband"
    strip_kw = {"strip": (rank, world)} if strip_mode else {}
    eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                           overlap_frames=args.overlap and not args.no_overlap and blend == "multiband" and not by_camera,
                           max_batch=batch, lens_composed=args.undistort and args.lens_composed, **strip_kw)
    in_flight = 1 if (by_camera or args.no_overlap) else (2 if args.overlap else max(1, min(args.in_flight, 8)))
    if in_flight == 2 and not args.overlap:
        in_flight = 3                   # (2 is the older two-stream form, multi-band only: --overlap asks for it)
    # the engine of the timed region: `in_flight` launch sequences in flight; `eng` stays the one-at-a-time engine of the
    # per-kernel timings and of the other legs
    eng_p = eng
    if in_flight > 1 and not args.overlap:
        eng_p = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                                 overlap_frames=True, overlap_depth=in_flight, max_batch=batch, lens_composed=args.undistort and args.lens_composed,
                                 **strip_kw)
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
    (cx, cy, cw, ch), _ = eng.acquire()
    if eng_p is not eng:
        eng_p.update(rig["scale"], rig["cameras"])
        eng_p.acquire()
    # the columns this rank writes: the whole canvas, or its strip.  A strip engine addresses a panorama from canvas column 0
    # and writes its own columns only: it is handed a strip-sized buffer shifted left by the strip's origin (`pptr`).
    x0, sw = 0, cw
    if strip_mode:
        sx_, _, sw, _ = eng.strip()
        x0 = sx_ - cx

    # frames resident in HBM: a few distinct sets, each rank its own (frame index offset by rank)
    dev = torch.device("cuda", local_rank)
    red_dev = None if smoke_backend else dev          # where the timing reductions' tensors live
    host_sets, dev_sets = [], []
    n_sets = max(args.frame_sets, batch)
    for s in range(n_sets):
        # (strip mode: every rank composes ITS COLUMNS of the same panoramas, so every rank holds the same frame sets)
        fr = rigmod.frames_for(rig, frame_index=s if strip_mode else sharding.global_frame_index(rank, world, s))
        host_sets.append(fr)
        dev_sets.append([torch.from_numpy(f).to(dev) for f in fr])
    # (--pano-row-pad: rows of cw * 3 + pad bytes, to time panorama strides that are not a multiple of four)
    pw = max(sw, 1)

    def make_panos():
        rows = [torch.empty((ch, pw * 3 + args.pano_row_pad), dtype=torch.uint8, device=dev) for _ in range(batch)]
        return rows, [p[:, :pw * 3].view(ch, pw, 3) if args.pano_row_pad == 0 else p for p in rows]

    def pptr(t):
        return t.data_ptr() - x0 * 3
    pano_rows, panos = make_panos()
    pano = panos[0]
    # calls in flight write different panoramas: a ring of `in_flight` sets of them
    pano_ring = [panos] + [make_panos()[1] for _ in range(in_flight - 1 if eng_p is not eng else 0)]
    # a real (non-null) stream, made torch's current one: torch.cuda.Event brackets the launches and RCCL's
    # point-to-point operations (camera-sharded mode) order against the same stream as the kernels
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    def tables(first):
        sets = [dev_sets[(first + k) % n_sets] for k in range(batch)]
        return [[t.data_ptr() for t in ds] for ds in sets], [[t.stride(0) for t in ds] for ds in sets]

    if by_camera:
        stitcher = sharding.ShardedStitcher(eng, rank, world, root=0, device=dev)

        def camera_call(ds):
            ptrs = {c: ds[c].data_ptr() for c in stitcher.mine}
            strides = {c: ds[c].stride(0) for c in stitcher.mine}
            return lambda: stitcher.compose(ptrs, strides, pano.data_ptr(), pano.stride(0), stream=stream)
        calls = [camera_call(ds) for ds in dev_sets]
    elif sw <= 0:
        calls = [lambda: None]           # (more ranks than 256-column strips: this rank has no columns; it still takes part in the timing)
    elif batch == 1:
        calls = [eng_p.make_device_call([t.data_ptr() for t in dev_sets[k % n_sets]], [t.stride(0) for t in dev_sets[k % n_sets]],
                                        pptr(pano_ring[k % len(pano_ring)][0]), pano.stride(0), stream=stream)
                 for k in range(n_sets * len(pano_ring))]
    else:
        calls = [eng_p.make_device_batch_call(*tables(k % n_sets), [pptr(p) for p in pano_ring[k % len(pano_ring)]], pano.stride(0),
                                              stream=stream) for k in range(n_sets * len(pano_ring))]

    def barrier():
        if world > 1:
            dist.barrier()

    def run(n_calls, first=0):
        for i in range(n_calls):
            calls[(first + i) % len(calls)]()

    # size `reps` (launch sequences per step) from a short probe so that K steps last >= --min-time on every rank
    run(3)
    torch.cuda.synchronize()
    probe = 10
    t0 = time.perf_counter()
    run(probe)
    torch.cuda.synchronize()
    t_seq = (time.perf_counter() - t0) / probe
    t_seq = sharding.max_over_ranks([t_seq], device=red_dev)[0]
    # (sized 25 % long: the probe runs right after the first launches and reads slower than the steady state)
    reps = max(1, int(math.ceil(1.25 * args.min_time / max(args.steps, 1) / max(t_seq, 1e-6))))

    for i in range(args.warmup):
        run(reps, i * reps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    clocks = ClockSampler()
    with clocks:                    # (a sysfs reader on its own thread: what clock and power the timed steps ran at)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            run(reps, i * reps)
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    wall, dev_ms = sharding.max_over_ranks([wall, dev_ms], device=red_dev)     # slowest rank sets the job time

    workload = "%s: %dx%dx%d BGR8 streams, %s warp, gain_blocks, %s blend (5 bands), canvas %dx%d" % (
        args.config, n_cam, rig["size"][0], rig["size"][1], rig["warp"], blend, cw, ch)
    alg_bytes = rigmod.algorithmic_bytes(rig, (cw, ch))
    out = None
    if rank == 0 and by_camera:
        seqs = args.steps * reps                      # one panorama per launch sequence for the whole job
        out = {
            "metric": METRIC, "value": round(seqs / wall, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall * 1000.0 / args.steps, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8/s16 fixed-point (f32 weights)", "data": "synthetic",
            "config": {"workload": workload, "frames_resident_in_hbm": True, "frame_sets_per_step": reps,
                       "parallelism": "cameras sharded over %d rank(s), pyramid payload sent to rank 0 (RCCL point-to-point)" % world},
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes * seqs / wall / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes * seqs / wall / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                         "scope": "whole job (all ranks' kernels + exchange per panorama)"},
        }
    elif rank == 0:
        # per-kernel durations: HIP events between launches on the compose stream, averaged over a few sequences
        reps_prof = 10
        acc, order = {}, []
        for i in range(reps_prof if sw > 0 else 0):
            # (a plain call queued right in front: the profiled sequence's first event is then recorded behind running kernels and
            # its first kernel's figure is the kernel, not the launch latency into an idle queue -- 30 us at one frame set per call)
            eng.compose_device_batch(*tables((i + 1) % n_sets), [pptr(p) for p in panos], pano.stride(0), stream=stream)
            st = eng.profile_stages_batch(*tables(i % n_sets), [pptr(p) for p in panos], pano.stride(0),
                                          stream=stream)
            for name, ms in st:
                if name not in acc:
                    acc[name] = [0.0, 0]
                    order.append(name)
                acc[name][0] += ms
                acc[name][1] += 1
        kernels = [{"name": k, "launches_per_sequence": acc[k][1] // reps_prof, "avg_ms": round(acc[k][0] / acc[k][1], 5),
                    "ms_per_frame": round(acc[k][0] / reps_prof / batch, 5)} for k in order]
        dominant = max(kernels, key=lambda k: k["ms_per_frame"]) if kernels else None
        seqs = args.steps * reps
        ms_per_step = wall * 1000.0 / args.steps
        dev_ms_per_seq = dev_ms / seqs
        # (`traffic` is never measured by this run: it is read from profiles/traffic.json -- the rocprofv3 PMC passes of
        # tools/profile_gpu.sh -- and only while that file was produced from the kernel sources being run)
        traffic, traffic_note, tj = None, "profiles/traffic.json absent", {}
        wl = args.config + (("+lens-composed" if args.lens_composed else "+undistort") if args.undistort else "")
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("kernel_source_sha16") == kernel_source_hash():
                ent = tj["entries"]["%s/%d" % (wl, batch)]
                traffic = ent["bytes_per_frame"] * batch
                traffic_note = "from profiles/traffic.json, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (%s), same kernel sources (sha %s)" % (
                    ent.get("source", ""), tj.get("kernel_source_sha16"))
            else:
                traffic_note = "profiles/traffic.json predates the current kernel sources (re-run tools/profile_gpu.sh)"
                tj = {}
        except (OSError, KeyError, ValueError):
            pass
        # one launch of a kernel processes `batch` frame sets: algorithmic bytes per launch = batch x frame.
        # Whole-step figure: all launches of a sequence together move those bytes once.
        # (strip mode: the N ranks compose the SAME panoramas, a strip each -- `achieved` is the whole job's rate on the whole
        # job's algorithmic bytes, `peak` the N GPUs' HBM)
        achieved = alg_bytes * batch / (dev_ms_per_seq * 1e-3) / 1e9
        peak = HBM_PEAK_GBS * (world if strip_mode else 1)
        out = {
            "metric": METRIC,
            "value": round(seqs * batch / wall if strip_mode else sharding.aggregate_throughput(world, seqs * batch, wall), 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "strong" if strip_mode else "weak",
            "vs_baseline": None,
            "dtype": "u8/s16 fixed-point (f32 weights)",
            "data": "synthetic",
            "config": {"workload": workload, "frames_resident_in_hbm": True, "frame_sets_per_step": batch * reps,
                       "launch_sequences_per_step": reps, "frame_sets_per_launch": batch,
                       "launch_sequences_in_flight": in_flight,
                       "lens_undistort": ("composed into the warp map" if args.lens_composed else "pre-pass") if args.undistort else False,
                       "timed_region_s": round(wall, 4),
                       "parallelism": ("canvas cut into %d column strip(s), rank r composes strip r of every panorama (restricted work lists, "
                                       "pixels identical to the whole-canvas run), strips left in place: no data-path collective" % world)
                                      if strip_mode else "frames sharded over %d rank(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
                         "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_frame": alg_bytes,
                         "algorithmic_bytes_per_launch": alg_bytes * batch,
                         "device_ms_per_frame": round(dev_ms_per_seq / batch, 5),
                         "scope": "whole launch sequence (%d launches over %d frame sets), %d sequence(s) in flight; `kernels` are "
                                  "timed one sequence at a time (events between the launches) -- under overlap a kernel's duration "
                                  "is not its own; the one-at-a-time rate is in `serial`"
                                  % (sum(k["launches_per_sequence"] for k in kernels), batch, in_flight),
                         "dominant_kernel": dominant, "kernels": kernels},
        }
        # ---- the ceiling that binds this path: instruction issue, not HBM.  Wave-instruction counts per
        # frame set from the rocprofv3 PMC passes behind profiles/traffic.json (same kernel sources), cycles per VALU
        # wave-instruction from tools/ubench/valu_rate.hip (this path's mix: ~3.0), the clock the timed steps actually ran at.
        clk = clocks.report()
        ent = tj.get("entries", {}).get("%s/%d" % (wl, batch), {}) if world == 1 or not strip_mode else {}
        if "valu_winst_per_frame" in ent:
            sclk_ghz = (clk or {}).get("sclk_mhz", {}).get("median", 2400.0) / 1e3
            n_simd, n_cu, cyc_valu = 1024, 256, 3.0
            valu_us = ent["valu_winst_per_frame"] * cyc_valu / (n_simd * sclk_ghz * 1e3)
            lds_us = ent.get("lds_active_cycles_per_frame", 0) / (n_cu * sclk_ghz * 1e3)
            out["roofline"]["issue"] = {
                "bound": "instruction issue (VALU + LDS share a SIMD's issue and register ports; vector-memory instructions ride on top)",
                "valu_winst_per_frame": ent["valu_winst_per_frame"], "lds_winst_per_frame": ent.get("lds_winst_per_frame"),
                "vmem_winst_per_frame": (ent.get("vmem_rd_winst_per_frame", 0) + ent.get("vmem_wr_winst_per_frame", 0)) or None,
                "salu_winst_per_frame": ent.get("salu_winst_per_frame"),
                "lds_active_cycles_per_frame": ent.get("lds_active_cycles_per_frame"),
                "lds_conflict_cycles_per_frame": ent.get("lds_conflict_cycles_per_frame"),
                "cycles_per_valu_winst": cyc_valu, "simds": n_simd, "cus": n_cu, "sclk_ghz": round(sclk_ghz, 3),
                "valu_issue_us_per_frame": round(valu_us, 2), "lds_issue_us_per_frame": round(lds_us, 2),
                "issue_bound_us_per_frame": round(valu_us + lds_us, 2),
                "measured_us_per_frame": round(dev_ms_per_seq / batch * 1e3, 2),
                "frac_of_issue_bound": round((valu_us + lds_us) / (dev_ms_per_seq / batch * 1e3), 4),
                "note": "counters: rocprofv3 --pmc passes of tools/profile_gpu.sh (SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_VMEM_* / SQ_LDS_IDX_ACTIVE, "
                        "one sequence at a time), not measured in this run; cycles per instruction: tools/ubench/valu_rate.hip"}
        if clk:
            out["roofline"]["clocks_under_load"] = clk
        if smoke_backend:
            out["smoke"] = "MST_BENCH_SMOKE_BACKEND=%s: %d ranks on %d visible device(s) -- a code-path smoke test, the numbers mean nothing" % (
                smoke_backend, world, torch.cuda.device_count())
        if batch > 1 and not args.no_latency_leg:
            # latency mode on the same engine: one frame set per launch sequence, as the reference's StitchingAll()
            one = [eng.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pano.data_ptr(),
                                        pano.stride(0), stream=stream) for ds in dev_sets]
            for i in range(20):
                one[i % len(one)]()
            torch.cuda.synchronize()
            n1 = max(50, int(0.4 / max(t_seq / batch, 1e-6)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n1):
                one[i % len(one)]()
            e1.record()
            torch.cuda.synchronize()
            ms1 = e0.elapsed_time(e1) / n1
            out["latency_mode"] = {"frame_sets_per_launch": 1, "frames": n1, "ms_per_frame": round(ms1, 5),
                                   "frames_per_s": round(1000.0 / ms1, 2),
                                   "roofline_frac": round(alg_bytes / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                   "traffic": tj.get("entries", {}).get("%s/1" % wl, {}).get("bytes_per_frame")}

    # ---- the same launch sequences strictly one after the other (`eng`): the rate the per-kernel durations and the
    # rocprofv3 profiles (tools/profile_gpu.sh runs this file with --in-flight 1) add up to.  Beside `value`, not `value`.
    if rank == 0 and world == 1 and not by_camera and eng_p is not eng:
        if batch == 1:
            calls_s = [eng.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pano.data_ptr(), pano.stride(0), stream=stream)
                       for ds in dev_sets]
        else:
            calls_s = [eng.make_device_batch_call(*tables(first), [p.data_ptr() for p in panos], pano.stride(0), stream=stream)
                       for first in range(n_sets)]
        for i in range(10):
            calls_s[i % len(calls_s)]()
        torch.cuda.synchronize()
        n2 = max(20, int(0.5 / max(t_seq, 1e-6)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n2):
            calls_s[i % len(calls_s)]()
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / n2
        out["serial"] = {"launch_sequences_in_flight": 1, "frame_sets_per_launch": batch, "frames": n2 * batch,
                         "frames_per_s": round(1000.0 * batch / ms2, 2), "device_ms_per_frame": round(ms2 / batch, 5),
                         "roofline_frac": round(alg_bytes * batch / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    # the reference's own call shape -- one frame set in, one panorama out per call -- with `in_flight` such calls in flight
    if rank == 0 and world == 1 and not by_camera and eng_p is not eng and "latency_mode" in out:
        one2 = [eng_p.make_device_call([t.data_ptr() for t in dev_sets[k % n_sets]], [t.stride(0) for t in dev_sets[k % n_sets]],
                                       pano_ring[k % len(pano_ring)][0].data_ptr(), pano.stride(0), stream=stream)
                for k in range(n_sets * len(pano_ring))]
        for i in range(20):
            one2[i % len(one2)]()
        torch.cuda.synchronize()
        n3 = max(50, int(0.4 / max(t_seq / batch, 1e-6)))
        t0 = time.perf_counter()
        for i in range(n3):
            one2[i % len(one2)]()
        torch.cuda.synchronize()
        w3 = time.perf_counter() - t0
        out["latency_mode"]["pipelined_calls"] = {"calls_in_flight": in_flight, "frames": n3, "frames_per_s": round(n3 / w3, 2),
                                                  "roofline_frac": round(alg_bytes * n3 / w3 / 1e9 / HBM_PEAK_GBS, 5)}

    # ---- the one-frame-set call replayed as a hipGraph (MST_FLAG_GRAPH_REPLAY): same kernels, one hipGraphLaunch per call ----
    if rank == 0 and world == 1 and not by_camera and "latency_mode" in out:
        try:
            eng3 = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                                    graph_replay=True, max_batch=batch)
            eng3.update(rig["scale"], rig["cameras"])
            eng3.acquire()
            ring = [(ds, panos[k % len(panos)]) for k, ds in enumerate(dev_sets)]

            def enqueue_rate(engine, n):
                for i in range(3 * len(ring)):                       # (first turn: capture)
                    ds, pn = ring[i % len(ring)]
                    engine.compose_device([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pn.data_ptr(), pn.stride(0), stream=stream)
                torch.cuda.synchronize()
                calls = [engine.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pn.data_ptr(), pn.stride(0), stream=stream)
                         for ds, pn in ring]
                t_host = 0.0
                for rep in range(10):                                # the host's share: a dozen calls into an empty queue
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(12):
                        calls[i % len(calls)]()
                    t_host += (time.perf_counter() - t0) / 12
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    calls[i % len(calls)]()
                torch.cuda.synchronize()
                return t_host / 10, (time.perf_counter() - t0) / n
            n4 = max(50, int(0.3 / max(t_seq / batch, 1e-6)))
            h_plain, w_plain = enqueue_rate(eng, n4)
            h_graph, w_graph = enqueue_rate(eng3, n4)
            out["latency_mode"]["graph_replay"] = {"frames": n4, "frames_per_s": round(1.0 / w_graph, 2),
                                                   "host_enqueue_us_per_call": round(h_graph * 1e6, 2),
                                                   "host_enqueue_us_per_call_kernel_by_kernel": round(h_plain * 1e6, 2),
                                                   "frames_per_s_kernel_by_kernel": round(1.0 / w_plain, 2)}
            eng3.close()
        except Exception as e:          # noqa: BLE001
            out["latency_mode"]["graph_replay"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- host buffers: the PCIe-inclusive rate (never `value`): serial mst_compose vs the three-stage pipeline ----
    if rank == 0 and world == 1 and not by_camera and not args.no_latency_leg and blend == "multiband":
        try:
            out["host_buffers"] = host_buffer_leg(eng, host_sets, cw, ch)
        except Exception as e:          # noqa: BLE001
            out["host_buffers"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the same pipeline fed NV12 frames (MST_FLAG_FRAMES_NV12: half the bytes over PCIe, converted on the device)
        try:
            import numpy as np
            eng_nv = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank, frames_nv12=True)
            eng_nv.update(rig["scale"], rig["cameras"])
            fw, fh = rig["size"]
            rng = np.random.default_rng(7)
            nv_sets = [[rng.integers(16, 236, (fh * 3 // 2, fw)).astype(np.uint8) for _ in range(n_cam)] for _ in range(2)]
            hb = host_buffer_leg(eng_nv, nv_sets, cw, ch, serial=False)
            out["host_buffers"]["nv12_pipeline_fps"] = hb["pipeline_fps"]
            out["host_buffers"]["nv12_mb_in_per_frame"] = hb["mb_in_per_frame"]
            eng_nv.close()
        except Exception as e:          # noqa: BLE001
            out["host_buffers"]["nv12_error"] = "%s: %s" % (type(e).__name__, e)
        # NV12 frames resident in HBM (no PCIe): what the conversion pre-pass costs the compose call itself
        try:
            import numpy as np
            eng_nv = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                                      frames_nv12=True, max_batch=batch)
            eng_nv.update(rig["scale"], rig["cameras"])
            eng_nv.acquire()
            fw, fh = rig["size"]
            rng = np.random.default_rng(11)
            nv_dev = [[torch.from_numpy(rng.integers(16, 236, (fh * 3 // 2, fw)).astype(np.uint8)).to(dev) for _ in range(n_cam)] for _ in range(batch)]
            pn = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)]
            call = eng_nv.make_device_batch_call([[t.data_ptr() for t in fs] for fs in nv_dev], [[t.stride(0) for t in fs] for fs in nv_dev],
                                                 [p.data_ptr() for p in pn], pn[0].stride(0), stream=stream)
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            nn = max(20, int(0.4 / max(t_seq, 1e-6)))
            t0 = time.perf_counter()
            for _ in range(nn):
                call()
            torch.cuda.synchronize()
            fps_nv = nn * batch / (time.perf_counter() - t0)
            out["nv12_resident"] = {"frames_per_s": round(fps_nv, 1), "of_bgr_rate": round(fps_nv / out.get("serial", {}).get("frames_per_s", out["value"]), 4),
                                    "note": "frames resident in HBM as NV12 (1.5 bytes a pixel), k_nv12_to_bgr ahead of every launch sequence"}
            eng_nv.close()
            del nv_dev, pn
        except Exception as e:          # noqa: BLE001
            out["nv12_resident"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # SURVEY 8 f3 end to end: one capture process per camera -> shared memory (the reference's DatePacket hand-off) ->
        # HBM -> compose -> panorama in page-locked host memory, BGR and NV12 packets
        try:
            out["host_buffers"]["ingest"] = ingest_leg(rig, blend, local_rank, cw, ch)
        except Exception as e:          # noqa: BLE001
            out["host_buffers"]["ingest"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- N > 1 legs: each under its own watchdog.  Every rank: a hung exchange must not take the headline (or the launcher)
    # with it.  A timer is armed per leg and says which leg it caught; none is running while the collective-free work
    # below (configs, CPU baseline) executes; the last one covers the final barrier and the teardown of the process group
    # (when a leg fails on one rank only, its peers sit in the exchange and this rank would otherwise wait for them
    # there).  A run that ends through a watchdog has still printed the headline, and exits non-zero.
    watchdog = [None]
    ranks_error = None

    def arm(leg, seconds=240.0):
        if watchdog[0] is not None:
            watchdog[0].cancel()
            watchdog[0] = None
        if leg is None:
            return

        def give_up():
            if rank == 0 and not printed:
                out.setdefault(leg, {})
                if isinstance(out[leg], dict):
                    out[leg]["error"] = "timed out after %d s in this leg" % int(seconds)
                else:
                    out[leg] = {"error": "timed out after %d s in this leg" % int(seconds)}
                print(json.dumps(out), flush=True)
            # a SECONDARY leg that hangs (an exchange that never completes on this node) must not cost the headline its exit code: the
            # line has been printed with the leg's error inside it.  Only a job that did not span its N ranks fails the run.
            os._exit(3 if leg == "ranks_seen" else 0)
        watchdog[0] = threading.Timer(seconds if rank == 0 else seconds + 10.0, give_up)
        watchdog[0].daemon = True
        watchdog[0].start()

    # ---- N > 1: which ranks and devices RCCL actually spans (the driver's "did it see N ranks" check reads this) ----
    if world > 1:
        arm("ranks_seen", 120.0)
        props = torch.cuda.get_device_properties(local_rank)
        ident = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "device": props.name,
                 "uuid": str(getattr(props, "uuid", "")), "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))}
        seen = [None] * world
        try:
            dist.all_gather_object(seen, ident)
        except Exception as e:          # noqa: BLE001
            seen = [{"error": "%s: %s" % (type(e).__name__, e)}]
        n_ranks = len([r for r in seen if isinstance(r, dict) and "rank" in r])
        # (distinct devices: by the index each rank made current -- one node, LOCAL_RANK = device index -- together with whatever the
        # runtime reports as uuid / PCI address; a stack that reports neither still counts N devices for N device indices)
        n_devs = len({((r or {}).get("uuid"), (r or {}).get("pci"), (r or {}).get("device_index")) for r in seen if isinstance(r, dict)})
        if rank == 0:
            out["ranks_seen"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": seen, "distinct_devices": n_devs}
        # a job that does not span N ranks on N distinct GPUs is not the N-GPU measurement: say so in the line and exit non-zero
        # (the one-GPU smoke hook, where the ranks share the visible devices on purpose, is exempt from the device count)
        if n_ranks != world or (args.gpus > 1 and dist.get_world_size() != args.gpus) or (not smoke_backend and n_devs != world):
            ranks_error = "ranks_seen: %d rank(s) on %d distinct device(s), expected %d (--gpus %d)" % (n_ranks, n_devs, world, args.gpus)
            if rank == 0:
                out["error"] = ranks_error
                out["value"] = None
    # ---- N > 1: the north-star split measured beside the headline (C4: 8 cameras, camera i on rank i % N) ----
    if (world > 1 or args.camera_leg) and not by_camera and not args.no_camera_leg:
        arm("camera_sharded")
        try:
            leg = camera_sharded_leg(args, rank, world, local_rank, dev, stream)
        except Exception as e:          # noqa: BLE001  -- the headline must survive a failure of the secondary leg
            leg = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            out["camera_sharded"] = leg
    # ---- N > 1: the canvas cut into N column strips, rank r composes strip r, strips concatenated on rank 0: the headline
    # workload (C3) and the north star's 8-camera rig (C4) ----
    if (world > 1 or args.strip_leg) and not by_camera and not args.no_camera_leg:
        for cname in ("C3", "C4"):
            arm("strip_sharded")
            try:
                leg = strip_sharded_leg(args, rank, world, local_rank, dev, stream, cname)
            except Exception as e:          # noqa: BLE001
                leg = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0:
                if cname == "C3":
                    out["strip_sharded"] = leg
                else:
                    out.setdefault("strip_sharded", {})[cname] = leg
    # ---- N > 1: the trivial mode beside the headline -- every GPU stitches its own frame sets, whole canvas, no exchange ("weak") ----
    if world > 1 and strip_mode:
        arm("replicas")
        try:
            leg = replicas_leg(rank, world, local_rank, dev, stream, rig, blend, dev_sets, batch, in_flight, args.undistort and args.lens_composed,
                               red_dev=red_dev)
        except Exception as e:          # noqa: BLE001
            leg = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            out["replicas"] = leg
    arm(None)       # nothing below talks to another rank until the final barrier

    # ---- the other single-GPU BASELINE configurations, through the same entry points (tools/bench_paths.py) ----
    if rank == 0 and world == 1 and not by_camera and not args.no_configs and args.config == "C3":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_paths
        eng.close()
        if eng_p is not eng:
            eng_p.close()
            eng_p = eng
        del dev_sets, pano_rows, panos, pano_ring, calls
        torch.cuda.empty_cache()
        cfgs = {}
        for nm in ("C2", "C5", "C3-LUT", "C5-LUT"):
            try:
                cfgs[nm] = bench_paths.measure(nm, batch=8, min_time=0.3, latency_leg=True)
                cfgs[nm].pop("name", None)
                cfgs[nm]["traffic"] = tj.get("entries", {}).get("%s/8" % nm, {}).get("bytes_per_frame") if rank == 0 else None
            except Exception as e:          # noqa: BLE001
                cfgs[nm] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["configs"] = cfgs

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not by_camera:
            out["cpu_baseline"] = cpu_baseline(rig, host_sets[0], blend, max_frames=args.cpu_frames, threads=args.cpu_threads)
        print(json.dumps(out), flush=True)
        printed = True

    if eng_p is not eng:
        eng_p.close()
    eng.close()
    if world > 1:
        arm("final_barrier", 120.0)
        dist.barrier()
        dist.destroy_process_group()
    arm(None)
    if ranks_error:
        sys.exit("bench.py: " + ranks_error)


def host_buffer_leg(eng, host_sets, cw, ch, serial=True):
    """Frames in host memory, panorama back to host memory (what the VC++ host hands over: cv::Mat data).
    serial: mst_compose per frame, pageable numpy buffers (H2D -> kernels -> D2H in series).
    pipeline: mst_pipeline_submit / wait, page-locked buffers, H2D(t+1) || kernels(t) || D2H(t-1)."""
    import numpy as np
    import stitchingvideo_amd as S
    n_ser = 30 if serial else 2
    t0 = time.perf_counter()
    for i in range(n_ser):
        eng.compose(host_sets[i % len(host_sets)], want_mask=False)
    serial = (time.perf_counter() - t0) / n_ser
    pinned = []
    for fr in host_sets:
        row = []
        for f in fr:
            pa = S.PinnedArray(f.shape)
            pa.array[...] = f
            row.append(pa)
        pinned.append(row)
    panos = [S.PinnedArray((ch, cw, 3)) for _ in range(3)]
    n_pipe = 300

    def loop(n):
        for t in range(n):
            if t >= 3:
                eng.pipeline_wait()
            eng.pipeline_submit([p.array for p in pinned[t % len(pinned)]], panos[t % 3].array)
        for _ in range(min(n, 3)):
            eng.pipeline_wait()
    loop(12)
    t0 = time.perf_counter()
    loop(n_pipe)
    piped = (time.perf_counter() - t0) / n_pipe
    frame_mb = sum(f.nbytes for f in host_sets[0]) / 1e6
    return {"serial_mst_compose_fps": round(1.0 / serial, 1), "serial_ms_per_frame": round(serial * 1e3, 3),
            "pipeline_fps": round(1.0 / piped, 1), "pipeline_ms_per_frame": round(piped * 1e3, 3),
            "mb_in_per_frame": round(frame_mb, 1), "mb_out_per_frame": round(ch * cw * 3 / 1e6, 1),
            "note": "PCIe-inclusive; serial = pageable buffers through mst_compose, pipeline = page-locked buffers "
                    "through mst_pipeline_submit/wait (3 slots)"}


def ingest_leg(rig, blend, local_rank, cw, ch, seconds=1.5):
    """N capture processes (python -m stitchingvideo_amd.ingest: synthetic frames, three distinct ones per camera, cycled) publish
    over POSIX shared memory in the reference's packet format; ingest.IngestStitcher pulls every camera's frame into HBM
    (all servers fill their segments concurrently, DMA out of the page-locked segments), composes, and copies the panorama
    into page-locked host memory -- three frame sets in flight.  PCIe-inclusive, never `value`."""
    import stitchingvideo_amd as S
    from stitchingvideo_amd import ingest
    n_cam = len(rig["cameras"])
    fw, fh = rig["size"]
    res = {}
    for fmt in ("bgr", "nv12"):
        names = ["/mst_bench_%s_%d_%d" % (fmt, os.getpid(), i) for i in range(n_cam)]
        # (the capture side copies into its segment with non-temporal stores: its consumer is a DMA engine -- mi355_ingest.h, MST_SHM_STREAM_COPY)
        procs = [subprocess.Popen([sys.executable, "-m", "stitchingvideo_amd.ingest", names[i], str(i), str(fw), str(fh), fmt, "3", "0"],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT, env=dict(os.environ, MST_SHM_STREAM_COPY="1"))
                 for i in range(n_cam)]
        eng = None
        try:
            for p in procs:
                if p.stdout.readline().strip() != "ready":
                    raise RuntimeError("capture process did not come up")
            # (the engine of the headline: calls in flight; IngestStitcher hands every call its frames-ready / panorama-free events)
            eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                                   frames_nv12=(fmt == "nv12"), overlap_frames=True, overlap_depth=4)
            eng.update(rig["scale"], rig["cameras"])
            def measure(st):
                def loop(n):
                    for t in range(n):
                        if t >= st.depth:
                            st.wait()
                        st.submit()
                    while st.completed < st.submitted:
                        st.wait()
                loop(6)
                t0 = time.perf_counter()
                loop(20)
                n = max(30, int(seconds / max((time.perf_counter() - t0) / 20, 1e-6)))
                t0 = time.perf_counter()
                loop(n)
                dt = time.perf_counter() - t0
                st.close()
                return n, dt
            bytes_in = sum((h * 3 // 2) * w if fmt == "nv12" else h * w * 3 for (w, h) in [c["size"] for c in rig["cameras"]])
            # (1) the native form: one mst_pipeline_submit_shm call per frame set on the engine's own three-stage host pipeline
            eng_n = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank, frames_nv12=(fmt == "nv12"))
            eng_n.update(rig["scale"], rig["cameras"])
            try:
                n, dt = measure(ingest.PipelineIngest(eng_n, names))
            finally:
                eng_n.close()
            res[fmt] = {"frames": n, "frames_per_s": round(n / dt, 1), "ms_per_frame": round(dt * 1e3 / n, 3),
                        "mb_in_per_frame": round(bytes_in / 1e6, 1), "mb_out_per_frame": round(ch * cw * 3 / 1e6, 1),
                        "pcie_in_gb_s": round(bytes_in * n / dt / 1e9, 2), "pcie_out_gb_s": round(ch * cw * 3 * n / dt / 1e9, 2),
                        "path": "mst_pipeline_submit_shm / mst_pipeline_wait (one native call per frame set)"}
            # (2) the same loop assembled in Python from torch streams + mst_compose_device_batch_ex (calls in flight, events)
            st = ingest.IngestStitcher(eng, names, [c["size"] for c in rig["cameras"]], device=local_rank)
            n, dt = measure(st)
            res[fmt]["python_streams_frames_per_s"] = round(n / dt, 1)
        finally:
            if eng is not None:
                eng.close()
            for p in procs:
                p.kill()
            for p in procs:
                p.wait(timeout=30)
            for nm in names:          # (killed servers leave their segments behind: they live in memory)
                try:
                    os.unlink("/dev/shm" + nm)
                except OSError:
                    pass
    res["note"] = ("%d capture processes -> shm (DatePacket) -> HBM -> compose (overlap_depth 4, mst_compose_device_batch_ex with "
                   "frames-ready / panorama-free events) -> page-locked panorama, 3 frame sets in flight; PCIe-inclusive" % n_cam)
    return res


def replicas_leg(rank, world, local_rank, dev, stream, rig, blend, dev_sets, batch, in_flight, lens_composed, seconds=0.6, red_dev="same"):
    """Every rank composes whole panoramas of the frame sets it holds (calls in flight as in the headline), nothing crosses ranks:
    the aggregate of N independent stitchers.  Weak scaling by construction; reported beside the strip-mode headline at N > 1."""
    import torch
    import torch.distributed as dist
    import stitchingvideo_amd as S
    from stitchingvideo_amd import sharding
    n_cam = len(rig["cameras"])
    eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend=blend, expos=rig["expos"], num_bands=5, device=local_rank,
                           overlap_frames=in_flight > 2, overlap_depth=in_flight if in_flight > 2 else 0, max_batch=batch, lens_composed=lens_composed)
    eng.update(rig["scale"], rig["cameras"])
    (_, _, cw, ch), _ = eng.acquire()
    ring = [[torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)] for _ in range(max(in_flight, 1))]
    n_sets = len(dev_sets)
    sets = [[dev_sets[(first + k) % n_sets] for k in range(batch)] for first in range(n_sets)]
    calls = [eng.make_device_batch_call([[t.data_ptr() for t in ds] for ds in sets[k % n_sets]], [[t.stride(0) for t in ds] for ds in sets[k % n_sets]],
                                        [p.data_ptr() for p in ring[k % len(ring)]], ring[0][0].stride(0), stream=stream)
             for k in range(n_sets * len(ring))]

    def run(n):
        for i in range(n):
            calls[i % len(calls)]()
    run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10)
    torch.cuda.synchronize()
    n = max(20, int(seconds / max((time.perf_counter() - t0) / 10, 1e-6)))
    red_dev = dev if red_dev == "same" else red_dev
    n = int(sharding.max_over_ranks([float(n)], device=red_dev)[0])
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = sharding.max_over_ranks([time.perf_counter() - t0], device=red_dev)[0]
    eng.close()
    return {"scaling": "weak", "frames_per_s": round(sharding.aggregate_throughput(world, n * batch, wall), 2), "frame_sets_per_launch": batch,
            "launch_sequences_in_flight": in_flight, "launch_sequences_per_rank": n,
            "parallelism": "frames sharded over %d rank(s): every rank stitches its own frame sets, no collective" % world}


def camera_sharded_leg(args_ns, rank, world, local_rank, dev, stream):
    """BASELINE C4 as specified: 8x1080p, camera i on GPU i % N, one exchange of the per-camera payload to rank 0
    per frame over RCCL (grouped point-to-point, every peer on its own xGMI link), rank 0 blends.  Strong scaling:
    the job is one panorama stream.  Runs on every rank; rank 0 returns the result object."""
    import torch
    import torch.distributed as dist
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    from stitchingvideo_amd import sharding
    cfg = rigmod.CONFIGS["C4"]
    rig = rigmod.build_rig(cfg)
    n_cam = len(rig["cameras"])
    eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, device=local_rank,
                           max_batch=2)                       # two pyramid sets: exchange of t overlaps stage of t + 1
    eng.update(rig["scale"], rig["cameras"])
    (_, _, cw, ch), _ = eng.acquire()
    st = sharding.ShardedStitcher(eng, rank, world, root=0, device=dev)
    st.refresh()
    sets = []
    for s in range(2):
        fr = rigmod.frames_for(rig, frame_index=s)
        sets.append({c: torch.from_numpy(fr[c]).to(dev) for c in st.mine})
    panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(2)]

    def args(t):
        ds = sets[t % len(sets)]
        return {c: x.data_ptr() for c, x in ds.items()}, {c: x.stride(0) for c, x in ds.items()}

    def run(n):
        # frame t's exchange is in flight while frame t + 1's per-camera kernels run; the canvas stage of t follows
        st.submit(0, *args(0), stream=stream)
        for t in range(1, n):
            st.submit(t, *args(t), stream=stream)
            st.finish(t - 1, panos[(t - 1) % 2].data_ptr(), panos[0].stride(0), stream=stream)
        st.finish(n - 1, panos[(n - 1) % 2].data_ptr(), panos[0].stride(0), stream=stream)

    run(10)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    n = max(10, args_ns.camera_leg_steps)
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = sharding.max_over_ranks([time.perf_counter() - t0], device=dev)[0]
    payload = st.payload_bytes()
    eng.close()
    alg = rigmod.algorithmic_bytes(rig, (cw, ch))
    return {"workload": "C4: %dx%dx%d BGR8 streams, spherical warp, gain_blocks, multiband blend (5 bands), canvas %dx%d"
                        % (n_cam, rig["size"][0], rig["size"][1], cw, ch),
            "parallelism": "camera i on rank i %% %d, packed tile payload to rank 0 over RCCL (one grouped point-to-point "
                           "exchange per frame, overlapped with the next frame's per-camera kernels), rank 0 blends" % world,
            "scaling": "strong", "frames": n, "frames_per_s": round(n / wall, 2), "ms_per_frame": round(wall * 1e3 / n, 4),
            "payload_bytes_per_frame_to_root": payload,
            "roofline_frac": round(alg * n / wall / 1e9 / HBM_PEAK_GBS, 5)}


def strip_sharded_leg(args_ns, rank, world, local_rank, dev, stream, config="C3"):
    """The canvas split over the GPUs (second form) on the headline workload C3 (and on C4): rank r's engine composes
    column strip r of `world` (mst_config.strip_index / strip_count: every kernel restricted to what the strip depends on,
    pixels identical to the whole-canvas run), 8 frame sets per launch sequence, and the strips of every panorama are
    concatenated on rank 0 by one grouped point-to-point step per launch sequence.  Every rank holds the frames of all
    cameras (a capture side would deliver each stream to the one or two GPUs whose strips it reaches).  Strong scaling."""
    import torch
    import torch.distributed as dist
    import stitchingvideo_amd as S
    from stitchingvideo_amd import rig as rigmod
    from stitchingvideo_amd import sharding
    cfg = rigmod.CONFIGS[config]
    rig = rigmod.build_rig(cfg)
    n_cam = len(rig["cameras"])
    batch = 8
    eng = S.PanoramaEngine(n_cam, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, device=local_rank,
                           max_batch=batch, strip=(rank, world))
    eng.update(rig["scale"], rig["cameras"])
    (cx, cy, cw, ch), _ = eng.acquire()
    sx, _, sw, _ = eng.strip()
    x0 = sx - cx
    sets = [[torch.from_numpy(f).to(dev) for f in rigmod.frames_for(rig, frame_index=s)] for s in range(batch)]
    strips = [torch.empty((ch, max(sw, 1), 3), dtype=torch.uint8, device=dev) for _ in range(batch)]
    panos = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(batch)] if rank == 0 else [None] * batch
    mine = torch.tensor([x0, sw], dtype=torch.int64, device=dev)
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rects = [(int(t[0]), int(t[1])) for t in allr]
    else:
        rects = [(x0, sw)]
    call = eng.make_device_batch_call([[t.data_ptr() for t in ds] for ds in sets], [[t.stride(0) for t in ds] for ds in sets],
                                      [b.data_ptr() - x0 * 3 for b in strips], strips[0].stride(0), stream=stream)

    def run(n, gather):
        for _ in range(n):
            if sw > 0:
                call()
            if gather and world > 1:
                for k in range(batch):
                    sharding.gather_strips_to_root(strips[k], rects, 0, rank, panos[k])
            elif gather:
                for k in range(batch):
                    panos[k][:, x0:x0 + sw] = strips[k]

    def timed(n, gather):
        run(3, gather)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n, gather)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return sharding.max_over_ranks([time.perf_counter() - t0], device=dev)[0]

    # (timed over at least half a second: a region of a few dozen milliseconds reads several per cent low)
    n = max(10, args_ns.camera_leg_steps)
    n = max(n, int(0.5 / max(timed(10, False) / 10, 1e-6)))
    t_compose = timed(n, False)
    t_total = timed(n, True)
    eng.close()
    # the same strips with four launch sequences in flight per strip engine (a strip's sequence is a small grid: most of what it
    # loses, it loses at launch boundaries -- DESIGN.md 4a' / 6), strips left in place
    t_flight = None
    if sw > 0:
        eng4 = S.PanoramaEngine(n_cam, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, device=local_rank,
                                max_batch=batch, strip=(rank, world), overlap_frames=True, overlap_depth=4)
        eng4.update(rig["scale"], rig["cameras"])
        eng4.acquire()
        ring = [strips] + [[torch.empty_like(b) for b in strips] for _ in range(3)]
        calls4 = [eng4.make_device_batch_call([[t.data_ptr() for t in ds] for ds in sets], [[t.stride(0) for t in ds] for ds in sets],
                                              [b.data_ptr() - x0 * 3 for b in r], strips[0].stride(0), stream=stream) for r in ring]
        state = {"i": 0}

        def call4():
            calls4[state["i"] % 4]()
            state["i"] += 1
        call = call4
        t_flight = timed(n, False)
        eng4.close()
    elif world > 1:
        t_flight = timed(n, False)          # (a rank without columns still takes part in the timing reduction)
    alg = rigmod.algorithmic_bytes(rig, (cw, ch))
    return {"workload": "%s: %dx%dx%d BGR8 streams, spherical warp, gain_blocks, multiband blend (5 bands), canvas %dx%d"
                        % (config, n_cam, rig["size"][0], rig["size"][1], cw, ch),
            "parallelism": "canvas cut into %d column strips, rank r composes strip r (restricted work lists), %d frame sets per launch, "
                           "strips concatenated on rank 0 (one grouped point-to-point step per launch sequence)" % (world, batch),
            "scaling": "strong", "strip_columns": rects,
            "frames_per_s": round(n * batch / t_total, 2), "frames_per_s_strips_left_in_place": round(n * batch / t_compose, 2),
            "frames_per_s_strips_left_in_place_4_sequences_in_flight": round(n * batch / t_flight, 2) if t_flight else None,
            "roofline_frac": round(alg * n * batch / t_total / 1e9 / HBM_PEAK_GBS, 5)}


if __name__ == "__main__":
    main()

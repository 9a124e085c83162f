"""One frame set per call, nothing beside it (the reference's StitchingAll() shape, DRV:724-770): the back-to-back rate of such
calls on one stream, the per-kernel durations of one call, and a truly lone call's latency (a synchronize after each).
    python tools/lone_call.py [C3]          -> one JSON line"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import stitchingvideo_amd as S                     # noqa: E402
from stitchingvideo_amd import rig as rigmod       # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    rig = rigmod.build_rig(rigmod.CONFIGS[name])
    n = len(rig["cameras"])
    dev = torch.device("cuda", 0)
    eng = S.PanoramaEngine(n, warp=rig["warp"], blend=rig["blend"], expos=rig["expos"], num_bands=5, max_batch=1)
    eng.update(rig["scale"], rig["cameras"])
    (_, _, cw, ch), _ = eng.acquire()
    sets = [[torch.from_numpy(f).to(dev) for f in rigmod.frames_for(rig, s)] for s in range(3)]
    pano = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    calls = [eng.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pano.data_ptr(), pano.stride(0), stream=stream.cuda_stream) for ds in sets]
    for i in range(30):
        calls[i % 3]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nrep = 3000
    e0.record()
    for i in range(nrep):
        calls[i % 3]()
    e1.record()
    torch.cuda.synchronize()
    back_to_back = e0.elapsed_time(e1) / nrep * 1e3
    t0 = time.perf_counter()
    for i in range(500):
        calls[i % 3]()
        torch.cuda.synchronize()
    lone = (time.perf_counter() - t0) / 500 * 1e6
    acc = {}
    for i in range(20):
        calls[(i + 1) % 3]()
        for nm, ms in eng.profile_stages_batch([[t.data_ptr() for t in sets[i % 3]]], [[t.stride(0) for t in sets[i % 3]]], [pano.data_ptr()], pano.stride(0), stream=stream.cuda_stream):
            acc.setdefault(nm, []).append(ms * 1e3)
    print(json.dumps({"config": name, "back_to_back_us_per_call": round(back_to_back, 2), "calls_per_s": round(1e6 / back_to_back, 1),
                      "lone_call_us_through_ctypes": round(lone, 1), "kernels_us": {k: round(sum(v) / 20, 2) for k, v in acc.items()}}))
    eng.close()


if __name__ == "__main__":
    main()

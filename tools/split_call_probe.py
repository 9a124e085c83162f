"""One frame set per call, SPLIT over the canvas: G strip engines (mst_config.strip_index / strip_count: every kernel restricted to what the
strip depends on, bit-exact), one stream each, all writing their own columns of ONE panorama.  The chains of the strips run beside each
other -- one strip's small launches and tails fill with another's large kernels -- which is what a lone call of the whole canvas cannot do.
    python tools/split_call_probe.py [C3] [G ...]          -> one line per G"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import stitchingvideo_amd as S                     # noqa: E402
from stitchingvideo_amd import rig as rigmod       # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    counts = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4]
    rig = rigmod.build_rig(rigmod.CONFIGS[name])
    n = len(rig["cameras"])
    dev = torch.device("cuda", 0)
    sets = [[torch.from_numpy(f).to(dev) for f in rigmod.frames_for(rig, s)] for s in range(3)]
    ref = S.PanoramaEngine(n, warp=rig["warp"], blend=rig["blend"], expos=rig["expos"], num_bands=5, max_batch=1)
    ref.update(rig["scale"], rig["cameras"])
    (_, _, cw, ch), _ = ref.acquire()
    want = torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev)
    ref.compose_device([t.data_ptr() for t in sets[1]], [t.stride(0) for t in sets[1]], want.data_ptr(), want.stride(0))
    torch.cuda.synchronize()
    want = want.cpu().numpy()
    ref.close()
    main_s = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_s)
    for g in counts:
        engs = [S.PanoramaEngine(n, warp=rig["warp"], blend=rig["blend"], expos=rig["expos"], num_bands=5, max_batch=1,
                                 **({"strip": (k, g)} if g > 1 else {})) for k in range(g)]
        for e in engs:
            e.update(rig["scale"], rig["cameras"])
            e.acquire()
        streams = [torch.cuda.Stream(device=dev) for _ in range(g)]
        pano = torch.zeros((ch, cw, 3), dtype=torch.uint8, device=dev)
        calls = [[e.make_device_call([t.data_ptr() for t in ds], [t.stride(0) for t in ds], pano.data_ptr(), pano.stride(0), stream=st.cuda_stream)
                  for e, st in zip(engs, streams)] for ds in sets]
        fork, joins = torch.cuda.Event(), [torch.cuda.Event() for _ in range(g)]

        def one(i):
            fork.record(main_s)
            for k in range(g):
                streams[k].wait_event(fork)
                calls[i % 3][k]()
                joins[k].record(streams[k])
            for k in range(g):
                main_s.wait_event(joins[k])

        one(1)
        torch.cuda.synchronize()
        exact = bool(np.array_equal(pano.cpu().numpy(), want))
        for i in range(30):
            one(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 2000
        e0.record(main_s)
        for i in range(nrep):
            one(i)
        e1.record(main_s)
        torch.cuda.synchronize()
        b2b = e0.elapsed_time(e1) / nrep * 1e3
        t0 = time.perf_counter()
        for i in range(300):
            one(i)
            torch.cuda.synchronize()
        lone = (time.perf_counter() - t0) / 300 * 1e6
        print(json.dumps({"config": name, "strips": g, "bit_exact": exact, "back_to_back_us_per_call": round(b2b, 2),
                          "lone_call_us_through_ctypes": round(lone, 1)}), flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()

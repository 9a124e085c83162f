"""The PCIe-inclusive legs of bench.py alone (host buffers through mst_pipeline_submit/wait; six capture processes -> shared memory ->
HBM -> compose -> page-locked panorama), for A/B runs of the ingest path: `python tools/ingest_bench.py [seconds] [prefetch 0|1]`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
import stitchingvideo_amd as S                      # noqa: E402
from stitchingvideo_amd import ingest, rig as rigmod   # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
mode = sys.argv[2] if len(sys.argv) > 2 else "2"     # 0: the round-5 behaviour (a camera is asked again only when the next set is pulled),
_init = ingest.IngestStitcher.__init__               # 1: prefetch, the call waits for the bus; 2: prefetch + asynchronous release (the default)
ingest.IngestStitcher.__init__ = lambda self, *a, **k: _init(self, *a, **dict(k, prefetch=mode != "0", async_release=mode == "2"))
rig = rigmod.build_rig(rigmod.CONFIGS["C3"])
eng = S.PanoramaEngine(6, warp=rig["warp"], blend="multiband", expos=rig["expos"], num_bands=5, device=0)
eng.update(rig["scale"], rig["cameras"])
(_, _, cw, ch), _ = eng.acquire()
host = [rigmod.frames_for(rig, frame_index=s) for s in range(3)]
out = {"host_buffers": bench.host_buffer_leg(eng, host, cw, ch, serial=False)}
eng.close()
out["ingest"] = bench.ingest_leg(rig, "multiband", 0, cw, ch, seconds=seconds)       # (frames_per_s: the native pipeline; python_streams_...: mode)
print(json.dumps(out))

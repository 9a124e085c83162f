import time, sys
sys.path.insert(0, '/root/repo')
import torch
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod
for name in ["C3", "C2"]:
    cfg = rigmod.CONFIGS[name]
    rig = rigmod.build_rig(cfg)
    eng = S.PanoramaEngine(len(rig["cameras"]), warp=rig["warp"], blend=cfg["blend"], expos=rig["expos"], max_batch=8)
    for rep in range(3):
        t0 = time.perf_counter(); eng.update(rig["scale"], rig["cameras"]); t1 = time.perf_counter()
        print(name, "update %.1f ms" % ((t1 - t0) * 1e3))
    eng.close()

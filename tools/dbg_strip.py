import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_edge_cases as T
import helpers as H
from stitchingvideo_amd import rig as rigmod
poses = [(-22.0, 18.0, 9.0), (3.0, -20.0, -12.0), (25.0, 7.0, 30.0), (60.0, 70.0, 5.0)]
rig = T.custom_rig("spherical", poses, (200, 150), 170.0)
frames = rigmod.frames_for(rig, 2)
eng = H.make_engine(rig, blend=None, num_bands=4, sharpness=0.02)
pano, mask = eng.compose(frames)
rect, opano, omask = H.oracle_compose(rig, frames, blend=None, num_bands=4, sharpness=0.02)
d = np.any(pano != opano, axis=2)
ys, xs = np.nonzero(d)
print("canvas", rect, "mismatch px", len(ys))
from collections import Counter
print("tiles", Counter(zip((ys // 64).tolist(), (xs // 128).tolist())))
print("x mod 128", sorted(Counter((xs % 128).tolist()).items()))
print("y mod 64", sorted(Counter((ys % 64).tolist()).items()))

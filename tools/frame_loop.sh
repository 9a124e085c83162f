#!/bin/bash
# GPU box: the C++ frame loop (tools/frame_loop_mi355.cpp) on the BASELINE rigs -- calibration files written by the Python
# rig builder, then lone calls / back to back / calls in flight from a C++ host.   tools/frame_loop.sh [depth]
cd ${GRAFT_REPO_ROOT:-.}
DEPTH=${1:-4}
make -C stitchingvideo_amd/csrc loop > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python3 - <<'PY'
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod, engine
for name in ("C3", "C2"):
    rig = rigmod.build_rig(rigmod.CONFIGS[name])
    open("/tmp/%s.cal" % name, "wb").write(engine.save_calibration(rig["scale"], rig["cameras"]))
PY
for spec in "C3 spherical multiband 1" "C3 spherical multiband 8" "C3 spherical no 1 --lut" "C2 plane feather 1" "C2 plane feather 8"; do
  set -- $spec
  ./tools/frame_loop_mi355 /tmp/$1.cal $2 $3 gain_blocks --depth $DEPTH --batch $4 $5 2>&1 | grep -v amdgpu.ids
done

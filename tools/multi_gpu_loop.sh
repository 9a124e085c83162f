#!/bin/bash
# GPU box: the multi-GPU frame loop as a C++ host (tools/multi_gpu_loop_mi355.cpp: C ABI + RCCL send/recv) on what one box offers --
# one rank (the exchange-free degenerate of both modes, checked byte for byte against the whole-canvas engine), and an attempt at two
# ranks on the one GPU (RCCL refuses two ranks of a communicator on one device: the message is recorded, nothing more can run here).
cd ${GRAFT_REPO_ROOT:-.}
make -C stitchingvideo_amd/csrc multi > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python3 - <<'PY'
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod, engine
for name in ("C3", "C4"):
    rig = rigmod.build_rig(rigmod.CONFIGS[name])
    open("/tmp/%s.cal" % name, "wb").write(engine.save_calibration(rig["scale"], rig["cameras"]))
PY
T=./tools/multi_gpu_loop_mi355
run() { timeout 300 "$@" 2>&1 | grep -v amdgpu.ids; }
run $T /tmp/C4.cal spherical gain_blocks --mode cameras --calls 300 --check
run $T /tmp/C3.cal spherical gain_blocks --mode strips --batch 1 --depth 4 --calls 400 --check
run $T /tmp/C3.cal spherical gain_blocks --mode strips --batch 8 --depth 4 --calls 200 --check
run $T /tmp/C3.cal spherical gain_blocks --mode strips --batch 8 --depth 4 --calls 200 --gather --check
echo "--- two ranks on one GPU (expected: refused by RCCL) ---"
run timeout 120 $T /tmp/C3.cal spherical gain_blocks --mode strips --batch 8 --depth 4 --calls 50 --gather --check --ranks 2 | tail -5

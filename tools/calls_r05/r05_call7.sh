#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
MST_COMPOSE_STATS=1 python -c "
import sys; sys.path.insert(0,'tests')
import stitchingvideo_amd as S
from stitchingvideo_amd import rig as rigmod
rig = rigmod.build_rig(rigmod.CONFIGS['C3'])
e = S.PanoramaEngine(6, warp=rig['warp'], blend='multiband', expos=rig['expos'], num_bands=5, max_batch=8); e.update(rig['scale'], rig['cameras']); e.close()" 2>&1 | grep "warp"
bash tools/variants.sh "p16:-DMST_A2_PITCH_ALIGN=16" "p1:-DMST_A2_PITCH_ALIGN=1" "p8:-DMST_A2_PITCH_ALIGN=8" "p16:-DMST_A2_PITCH_ALIGN=16" "p1:-DMST_A2_PITCH_ALIGN=1" "p8:-DMST_A2_PITCH_ALIGN=8" "nv12st:-DMST_ABLATE_NV12STAGE" "p16:-DMST_A2_PITCH_ALIGN=16" 2>&1 | tee gpurun_out/r05/ab_pitch.txt
bash tools/profile_gpu.sh r05_pitch quick 2>&1 | tail -30

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/variants.sh "th32:-DMST_A_TH0=32" "th16:-DMST_A_TH0=16 -DMST_A2_ROWS=2" "th32:-DMST_A_TH0=32" "th16:-DMST_A_TH0=16 -DMST_A2_ROWS=2" 2>&1 | tee gpurun_out/r05/ab_th0.txt
for v in th32 th16 th32 th16; do (cd /tmp/mst_var_$v && echo $v && python tools/lone_call.py C3 2>/dev/null && MST_WARP_STAGED_MIN=1 python tools/lone_call.py C3 2>/dev/null); done | tee -a gpurun_out/r05/ab_th0.txt

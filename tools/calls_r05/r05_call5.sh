#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for m in 0 2 0 2; do echo "MST_PYR_FUSED_MAX=$m"; MST_PYR_FUSED_MAX=$m timeout 120 python tools/lone_call.py C3 2>/dev/null; done | tee gpurun_out/r05/lone_pyr_fused.txt
for m in 0 8 0 8; do echo "MST_PYR_FUSED_MAX=$m (batches of 8)"; MST_PYR_FUSED_MAX=$m timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-latency-leg --min-time 0.4 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('in-flight', d['value'], '| serial', d.get('serial',{}).get('frames_per_s'), [(k['name'], k['ms_per_frame']) for k in d['roofline']['kernels']])"; done | tee -a gpurun_out/r05/lone_pyr_fused.txt
bash tools/frame_loop.sh 4 2>&1 | head -2 | tee -a gpurun_out/r05/lone_pyr_fused.txt

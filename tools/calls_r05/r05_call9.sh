#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', 'value', d['value'], d['roofline']['frac'], '| serial', d.get('serial',{}).get('frames_per_s'))"; }
for q in "" 2 4 8 16; do
  for f in 4 6 8; do
    if [ -z "$q" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-latency-leg --min-time 0.4 --in-flight $f 2>/dev/null | show "GPU_MAX_HW_QUEUES=${q:-default} in-flight=$f"
  done
done | tee gpurun_out/r05/hw_queues.txt
unset GPU_MAX_HW_QUEUES
for q in "" 8; do
  if [ -z "$q" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --min-time 0.4 --batch 1 --no-latency-leg 2>/dev/null | show "GPU_MAX_HW_QUEUES=${q:-default} one frame set per call, 4 in flight"
done | tee -a gpurun_out/r05/hw_queues.txt

#!/bin/bash
# round 5, call 17: calls in flight overlapped stage against stage (MST_CHAIN_MODE: 1 warp launches chained across calls, 2 canvas launches, 3 both), in-flight 4 and 6
mkdir -p gpurun_out/r05
for m in 0 1 2 3 0 3; do for d in 4 6; do
  MST_CHAIN_MODE=$m python bench.py --in-flight $d --no-cpu-baseline --no-configs --no-latency-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chain mode $m in flight $d: value', d['value'], 'frac', d['roofline']['frac'], 'serial', d['serial']['frames_per_s'])"
done; done | tee gpurun_out/r05/chain_mode.txt

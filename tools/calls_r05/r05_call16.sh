#!/bin/bash
# round 5, call 16: does `value` lean on the Infinity Cache holding the three synthetic frame sets the bench cycles through?  --frame-sets 3 / 16 / 48
mkdir -p gpurun_out/r05
for fs in 3 16 48 3 48; do
  python bench.py --frame-sets $fs --no-cpu-baseline --no-configs --no-latency-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frame sets $fs: value', d['value'], 'frac', d['roofline']['frac'], 'serial', d['serial']['frames_per_s'])"
done | tee gpurun_out/r05/frame_sets.txt

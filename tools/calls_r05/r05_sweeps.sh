#!/bin/bash
# final-build parity sweeps with seeds the suite does not use, and the strip-leg repeat
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for spec in "800 51001" "800 51002" "600 51003"; do python tests/sweeps/fuzz_parity.py $spec 2>&1 | tail -1; done | tee gpurun_out/r05/fuzz.txt
FUZZ_MAX=1400 python tests/sweeps/fuzz_parity.py 300 51004 2>&1 | tail -1 | tee -a gpurun_out/r05/fuzz.txt
python tests/sweeps/full_size_parity.py 2>&1 | tail -3 | tee -a gpurun_out/r05/fuzz.txt
for i in 1 2; do python bench.py --strip-leg --no-cpu-baseline --no-configs --no-latency-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('strip leg', d['strip_sharded']['frames_per_s_strips_left_in_place'], 'serial', d['serial']['frames_per_s'], 'value', d['value'])"; done | tee -a gpurun_out/r05/fuzz.txt

#!/bin/bash
# a second, larger parity sweep on the final build: 8000 random rigs with seeds nothing else uses, 600 of them at up to 1400-pixel frames
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
: > gpurun_out/r05/fuzz2.txt
for spec in "2000 52001" "2000 52002" "2000 52003" "1400 52004"; do timeout 900 python tests/sweeps/fuzz_parity.py $spec 2>&1 | tail -1; done | tee -a gpurun_out/r05/fuzz2.txt
FUZZ_MAX=1400 timeout 900 python tests/sweeps/fuzz_parity.py 600 52005 2>&1 | tail -1 | tee -a gpurun_out/r05/fuzz2.txt

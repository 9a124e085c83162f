#!/bin/bash
# round 5, GPU call 1: parity of the refactored level-0 code, packed vs dword level 0 (same box, same run, A B A B), clocks under load
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
(time python -m pytest tests -m gpu -x -q) 2>&1 | tail -4
VARIANT_PARITY="dword" bash tools/variants.sh "packed:-DMST_L0_PACKED=1" "dword:-DMST_L0_PACKED=0" "packed:-DMST_L0_PACKED=1" "dword:-DMST_L0_PACKED=0" "packed:-DMST_L0_PACKED=1" "dword:-DMST_L0_PACKED=0" 2>&1 | tee gpurun_out/r05/ab_l0.txt
python tools/clock_watch.py -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-latency-leg > gpurun_out/r05/clock_bench.json 2> gpurun_out/r05/clock_watch.txt
tail -1 gpurun_out/r05/clock_watch.txt
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40

#!/bin/bash
# instruction-issue counters for the multi-band kernels, a few per pass, each pass under `timeout`.
# Usage: tools/prof_sq.sh <tag> [bench args]   -> prints per-kernel averages per dispatch
TAG=${1:-sq}; shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
REPO=$PWD
BENCH="python3 $PWD/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-overlap $@"
cd /tmp
pass() { local name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
pass sqa SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES
pass sqb SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_CYCLES
pass sqc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU
pass grbm GRBM_GUI_ACTIVE
cd "$REPO"
python3 tools/summarize_counters.py "$OUT"

#!/bin/bash
# instruction-cache / issue counters for the multi-band kernels.  Usage: tools/prof_icache.sh <tag> [bench args]
TAG=${1:-ic}; shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 2 --min-time 0.2 --no-cpu-baseline --no-configs --no-latency-leg $@"
cd /tmp
pass() { local name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass ic2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass ic3 SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQC_DCACHE_MISSES
cd - > /dev/null
python3 tools/summarize_prof.py "$OUT" | grep -A14 -E "^k_(blend|warp_pyr|pyr_chain|compose)"

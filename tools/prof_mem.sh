#!/bin/bash
# memory-pipeline counters for the multi-band kernels (few counters per pass: TA/TCP blocks have 2-4 slots;
# every pass is wrapped in `timeout`: an over-subscribed pass aborts and would otherwise hang).
# Usage: tools/prof_mem.sh <tag> [bench args]
TAG=${1:-mem}; shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-overlap $@"
cd /tmp
pass() { local name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
pass ta1 TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp2 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
pass sq4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass grbm GRBM_GUI_ACTIVE
ls "$OUT"

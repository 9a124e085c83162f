#!/bin/bash
# rocprofv3 passes over one path of tools/bench_paths.py on the GPU box: tools/prof_paths.sh <tag> <path name> [bench_paths args]
TAG=$1; NAME=$2; shift 2
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $PWD/tools/bench_paths.py $NAME --in-flight 1 --min-time 0.1 --no-latency-leg $@"
echo "$BENCH" | sed "s|$PWD/||g" > "$OUT/command.txt"
REPO=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
pass() { local n=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- $BENCH > "$OUT/$n.log" 2>&1 || echo "pass $n failed"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" "$REPO/gpurun_out/traffic.json" > "$OUT/summary.txt"
cat "$OUT/summary.txt" | grep -v "^$" | head -80

#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench workload (batched launch
# sequences only: no latency leg, no CPU baseline), then the text summary and profiles/traffic.json-style totals.
# Usage: tools/profile_gpu.sh <tag> [quick] [bench args]      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-run}; shift
QUICK=0; if [ "${1:-}" = "quick" ]; then QUICK=1; shift; fi
EXTRA="$@"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --in-flight 1 --steps 10 --warmup 3 --min-time 0.1 --no-cpu-baseline --no-latency-leg --no-configs $EXTRA"
REPO=$PWD
echo "$BENCH" | sed "s|$PWD/||" > "$OUT/command.txt"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
pass() {   # name, counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
if [ $QUICK = 0 ]; then
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum
pass grbm GRBM_GUI_ACTIVE
fi
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" "$REPO/gpurun_out/traffic.json" > "$OUT/summary.txt"
grep -E "^k_(blend|warp_pyr|pyr_chain)|FETCH_SIZE|WRITE_SIZE|LDS_BANK|LDS_IDX|avg_us|^k_" "$OUT/summary.txt" | head -60

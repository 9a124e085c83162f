#!/bin/bash
# PMC passes over a micro-benchmark binary on the GPU box: tools/ubench/prof_pmc.sh <binary> <out dir>
# (counters in their own runs with --kernel-trace only; one table of per-kernel means at the end)
BIN=$1; OUT=$2; mkdir -p "$OUT"; export TMPDIR=/tmp; export SAMPLER_REPS=3
cd /tmp
pass() { local n=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- $BIN > "$OUT/$n.log" 2>&1 || echo "pass $n failed"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
pass ta GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
pass tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp2 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
pass mem FETCH_SIZE TCC_HIT_sum TCC_MISS_sum
pass wr WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections,re
root=sys.argv[1]
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+"/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
        d[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=sorted({c for v in d.values() for c in v})
for k,v in sorted(d.items()):
    print(k)
    for c in names:
        if c in v: print("   %-40s %16.0f" % (c, sum(v[c])/len(v[c])))
PY

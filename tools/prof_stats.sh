#!/bin/bash
# kernel-trace only: per-dispatch durations.  Usage: tools/prof_stats.sh <tag> [bench args]
TAG=${1:-run}; shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs $@"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
python3 - "$OUT/stats/stats_kernel_trace.csv" <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    m=re.search(r"(k_[a-z0-9_]+(<\w+>)?)",r["Kernel_Name"])
    if not m: continue
    key=(m.group(1), r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    d[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("%-18s %9s %5s %7s %5s %5s %6s %9s %9s"%("kernel","grid","wg","lds","vgpr","sgpr","calls","avg_us","min_us"))
for k,v in sorted(d.items()):
    print("%-18s %9s %5s %7s %5s %5s %6d %9.2f %9.2f"%(k+(len(v),sum(v)/len(v),min(v))))
PY

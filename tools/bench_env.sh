#!/bin/bash
# On the GPU box: time the default library under a list of environment settings, e.g.
#   tools/bench_env.sh "" "MST_WARP_STAGE=0" "MST_WARP_TABLE=1"
# Extra bench.py arguments come from BENCH_ARGS (default: --no-overlap).
BENCH_ARGS=${BENCH_ARGS:---no-overlap}
for setting in "$@"; do
  echo -n "[${setting:-default}] "
  env $setting MST_VERBOSE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $BENCH_ARGS 2> /tmp/bench_env.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(d['value'], [(k['name'],k['launches_per_step'],round(k['ms_per_frame']*1000,1)) for k in d['roofline']['kernels']])"
  grep -h "mi355stitch" /tmp/bench_env.err | sort | uniq | head -3
done

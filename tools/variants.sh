#!/bin/bash
# Build kernel variants (-D flags) in scratch copies on the GPU box and time each with bench.py.
# Usage: tools/variants.sh "name1:-DFOO=1 -DBAR=2" "name2:..." ...      (timing experiments; outputs may be wrong)
SRC=$PWD
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1',d['value'],d['roofline']['frac'],[ (k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']], d.get('latency_mode',{}).get('frames_per_s'))"; }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  W=/tmp/mst_var_$name; rm -rf $W; mkdir -p $W; cp -r $SRC/stitchingvideo_amd $SRC/include $SRC/tools $SRC/bench.py $SRC/tests $SRC/oracle $W/ 2>/dev/null
  (cd $W/stitchingvideo_amd/csrc && rm -rf build && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags" > $W/build.log 2>&1) || { echo "$name: build failed"; tail -5 $W/build.log; continue; }
  (cd $W && python3 bench.py --steps 20 --warmup 5 --in-flight 1 --no-cpu-baseline --no-configs --no-latency-leg --min-time 0.4 $BENCH_ARGS 2>/dev/null | show "$name")
  # VARIANT_PATHS="C2 C3-LUT": the other single-GPU paths through tools/bench_paths.py as well
  [ -n "$VARIANT_PATHS" ] && (cd $W && python3 tools/bench_paths.py $VARIANT_PATHS --no-latency-leg 2>/dev/null | python3 -c "
import sys,json
print('$name', ' '.join('%s %.4f' % (d['name'], d.get('roofline_frac', -1)) for d in map(json.loads, filter(lambda l: l.startswith('{'), sys.stdin))))")
done

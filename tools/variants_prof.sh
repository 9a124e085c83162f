#!/bin/bash
# Like tools/variants.sh, but reports per-kernel durations from a rocprofv3 kernel trace of each variant.
SRC=$PWD
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  W=/tmp/mst_var_$name; rm -rf $W; mkdir -p $W; cp -r $SRC/stitchingvideo_amd $SRC/include $SRC/tools $SRC/bench.py $SRC/tests $SRC/oracle $W/ 2>/dev/null
  (cd $W/stitchingvideo_amd/csrc && rm -rf build && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags" > $W/build.log 2>&1) || { echo "$name: build failed"; tail -5 $W/build.log; continue; }
  echo "== $name"
  (cd $W && bash tools/prof_stats.sh v_$name --no-latency-leg --min-time 0.1 $BENCH_ARGS 2>&1 | grep -E "^k_(blend|warp_pyr|pyr_chain)")
done

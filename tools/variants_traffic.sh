#!/bin/bash
# Like tools/variants.sh, but reports FETCH_SIZE / WRITE_SIZE per canvas kernel (rocprofv3 PMC passes) of each variant.
SRC=$PWD
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  W=/tmp/mst_var_$name; rm -rf $W; mkdir -p $W; cp -r $SRC/stitchingvideo_amd $SRC/include $SRC/tools $SRC/bench.py $SRC/tests $SRC/oracle $W/ 2>/dev/null
  (cd $W/stitchingvideo_amd/csrc && rm -rf build && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags" > $W/build.log 2>&1) || { echo "$name: build failed"; tail -5 $W/build.log; continue; }
  echo "== $name"
  (cd $W && bash tools/profile_gpu.sh v quick > /dev/null 2>&1; awk '/^k_blend_fine/,/^k_border/' gpurun_out/prof_v/summary.txt | grep -E "^k_blend_fine|FETCH_SIZE|WRITE_SIZE")
done

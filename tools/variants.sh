#!/bin/bash
# Build kernel-shape variants locally (before gpurun): tools/variants.sh "A32_C32 -DMST_A_TH0=32 -DMST_C_TH=32" ...
cd "$(dirname "$0")/../stitchingvideo_amd/csrc"
for spec in "$@"; do
  name=${spec%% *}; flags=${spec#* }
  make -s clean >/dev/null; make -s -j4 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter $flags" 2>&1 | grep -E "error" 
  cp ../libmi355stitch.so ../libmi355stitch_$name.so
done
make -s clean >/dev/null; make -s -j4 2>&1 | grep -E "error"

#!/bin/bash
# Build a -DMST_STAMPS copy of the library in a scratch directory and run tools/stamps.py against it (GPU box).
set -e
SRC=$PWD
W=/tmp/mst_stamps; rm -rf $W; mkdir -p $W; cp -r $SRC/stitchingvideo_amd $SRC/include $SRC/tools $W/
cd $W/stitchingvideo_amd/csrc && rm -rf build && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMST_STAMPS" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cd $W && python3 tools/stamps.py "$@"

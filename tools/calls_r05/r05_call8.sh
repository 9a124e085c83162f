#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
BENCH_ARGS="--config C2" VARIANT_PATHS="C2 C3-feather" bash tools/variants.sh "fw6:-DMST_COMPOSE_WAVES_FEATHER=6" "fw8:-DMST_COMPOSE_WAVES_FEATHER=8" "fw6:-DMST_COMPOSE_WAVES_FEATHER=6" "fw8:-DMST_COMPOSE_WAVES_FEATHER=8" "fw5:-DMST_COMPOSE_WAVES_FEATHER=5" 2>&1 | grep -v "^fw. in-flight" | tee gpurun_out/r05/ab_feather_waves.txt

#!/bin/bash
# round 5, call 13: the deep chain (k_pyr_deep) on lone calls -- parity first, then A/B of its shapes against the batched sequence
mkdir -p gpurun_out/r05
o=gpurun_out/r05/deep_chain.txt
: > $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $o
for v in "MST_DEEP_CHAIN=0" "MST_DEEP_T=32 MST_DEEP_NT=256" "MST_DEEP_T=32 MST_DEEP_NT=512" "MST_DEEP_T=64 MST_DEEP_NT=256" "MST_DEEP_T=64 MST_DEEP_NT=512" "MST_DEEP_CHAIN=0" "MST_DEEP_T=32 MST_DEEP_NT=256"; do
  echo "== $v" | tee -a $o
  env $v timeout 300 python tools/lone_call.py C3 2>&1 | tail -1 | tee -a $o
done
echo "== C5 off / on" | tee -a $o
MST_DEEP_CHAIN=0 timeout 300 python tools/lone_call.py C5 2>&1 | tail -1 | tee -a $o
timeout 300 python tools/lone_call.py C5 2>&1 | tail -1 | tee -a $o

#!/bin/bash
# round 5, call 14: kernel A's taps as one unaligned 8-byte LDS read per row (-DMST_A2_TAP_B64=1) against ds_read2_b32: parity, then A B A B
mkdir -p gpurun_out/r05
VARIANT_PARITY="b64" bash tools/variants.sh "base:-DMST_A2_TAP_B64=0" "b64:-DMST_A2_TAP_B64=1" "base2:-DMST_A2_TAP_B64=0" "b64_2:-DMST_A2_TAP_B64=1" 2>&1 | tee gpurun_out/r05/tap_b64.txt

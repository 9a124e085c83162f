#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for t in 256 512 256 512; do echo "MST_CC16_THREADS=$t"; MST_CC16_THREADS=$t python tools/lone_call.py C3 2>/dev/null; done | tee gpurun_out/r05/lone_cc16.txt
bash tools/variants.sh "base:-DMST_A_WAVES=7" "aw6:-DMST_A_WAVES=6" "aw5:-DMST_A_WAVES=5" 2>&1 | tee gpurun_out/r05/ab_awaves.txt
for v in base aw6 aw5; do (cd /tmp/mst_var_$v && echo $v && python tools/lone_call.py C3 2>/dev/null); done | tee -a gpurun_out/r05/ab_awaves.txt
echo "--- two ranks, one GPU ---"
timeout 120 ./tools/multi_gpu_loop_mi355 /tmp/C3.cal spherical gain_blocks --mode strips --batch 8 --depth 4 --calls 50 --gather --check --ranks 2 > gpurun_out/r05/two_ranks.txt 2>&1; echo "rc=$?" >> gpurun_out/r05/two_ranks.txt
grep -v "version\|Hostname\|Librccl" gpurun_out/r05/two_ranks.txt | tail -12

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/multi_gpu_loop.sh 2>&1 | tee gpurun_out/r05/multi_gpu_loop.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-latency-leg --strip-headline 2>/dev/null > gpurun_out/r05/bench_strip_headline.json
python -c "
import json; d=json.load(open('gpurun_out/r05/bench_strip_headline.json')); print('strip-headline at N=1:', d['value'], d['scaling'], d['roofline']['frac'], d.get('serial',{}).get('frames_per_s'), d['config']['parallelism'][:60])"
bash tools/frame_loop.sh 4 2>&1 | tee gpurun_out/r05/frame_loop.txt

#!/bin/bash
# round 5, GPU call 2: the whole -m gpu suite with the new stream-ordering tests, the clocks under the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_streams.py tests/test_ingest.py -m gpu -x -q 2>&1 | tail -15
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/clock_watch.py -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-latency-leg > gpurun_out/r05/clock_bench.json 2> gpurun_out/r05/clock_watch.txt
tail -1 gpurun_out/r05/clock_watch.txt

#!/bin/bash
# round 5, call 15: the -m gpu suite three times over on the final build (flakiness check of the stream-ordering tests), smoke(), default bench
mkdir -p gpurun_out/r05
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2; done | tee gpurun_out/r05/suite_x3.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r05/suite_x3.txt

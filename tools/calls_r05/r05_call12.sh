#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 1 1; do echo "forced, buffer_inv sc0 + plain loads"; MST_PYR_FUSED_FORCE=1 MST_PYR_FUSED_MAX=2 timeout 120 python tools/lone_call.py C3 2>/dev/null; done
MST_PYR_FUSED_FORCE=1 MST_PYR_FUSED_MAX=2 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_streams.py -m gpu -q 2>&1 | tail -3

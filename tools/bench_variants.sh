#!/bin/bash
# On the GPU box: time each prebuilt library variant (tools/variants.sh); PARITY=1 also runs the parity tests.
BENCH_ARGS=${BENCH_ARGS:---no-overlap}
for lib in stitchingvideo_amd/libmi355stitch_*.so; do
  echo -n "$(basename $lib): "
  if [ -n "$PARITY" ]; then
    MST_LIB=$PWD/$lib python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or C3-4" 2>&1 | tail -1 | tr '\n' ' '
  fi
  MST_LIB=$PWD/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline $BENCH_ARGS | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(d['value'], [(k['name'],k['launches_per_step'],round(k['ms_per_frame']*1000,1)) for k in d['roofline']['kernels']])"
done

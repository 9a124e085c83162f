cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
python -m pytest tests -m gpu -x -q -k "undistort" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-latency-leg --steps 20 --warmup 5 --min-time 0.5 > gpurun_out/r2f/bench_plain.json 2>/dev/null
python bench.py --undistort --no-cpu-baseline --no-latency-leg --steps 20 --warmup 5 --min-time 0.5 > gpurun_out/r2f/bench_undistort.json 2>/dev/null
python -c "
import json
for f in ('bench_plain','bench_undistort'):
    d=json.load(open('gpurun_out/r2f/%s.json'%f)); print(f, d['value'], [(k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']])"

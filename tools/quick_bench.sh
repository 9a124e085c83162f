python tools/bench_paths.py C3-LUT C5-LUT C2 C3-paste C3-feather C2-LUT --no-latency-leg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['name'], d.get('roofline_frac'), d.get('error',''))"
python bench.py --no-cpu-baseline --no-configs --no-latency-leg --steps 20 --warmup 5 --min-time 0.4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C3', d['value'], d['roofline']['frac'], [(k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']])"
python bench.py --undistort --no-cpu-baseline --no-configs --no-latency-leg --steps 20 --warmup 5 --min-time 0.4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C3-undistort', d['value'], d['roofline']['frac'], [(k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']][:2])"

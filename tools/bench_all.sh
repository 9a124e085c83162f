#!/bin/bash
for c in C2 C3 C4 C5; do
  python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('$c', d['value'],'fps', d['ms_per_step'],'ms', 'alg MB %.1f'%(r['algorithmic_bytes_per_frame']/1e6), 'achieved GB/s', r['achieved'], 'frac', r['frac'], [(k['name'],k['launches_per_frame'],round(k['ms_per_frame']*1000,1)) for k in r['kernels']])"
done

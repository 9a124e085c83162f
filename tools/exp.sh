for e in 0 1 2 4 8 16 31 32 64 96; do
  echo -n "EXP=$e: "; MST_EXP=$e python bench.py --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(d['value'], [(k['name'],round(k['ms_per_frame']*1000,1)) for k in d['roofline']['kernels']])"
done

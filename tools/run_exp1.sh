cd $GRAFT_REPO_ROOT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1',d['value'],d['roofline']['frac'],[ (k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']])"; }
for b in 1 2 4 8; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-leg --min-time 0.3 --batch $b 2>/dev/null | show batch$b
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-leg --min-time 0.3 --batch 8 --overlap 2>/dev/null | show overlap8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-leg --min-time 0.3 --batch 4 --overlap 2>/dev/null | show overlap4
for b in 1 2; do
bash tools/prof_stats.sh exp1_b$b --no-latency-leg --min-time 0.1 --batch $b 2>&1 | grep -E "^kernel|^k_(blend|warp_pyr|pyr_chain)"
done

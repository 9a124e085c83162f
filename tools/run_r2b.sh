cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2b/pytest.log 2>&1
tail -4 gpurun_out/r2b/pytest.log
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1',d['value'],d['roofline']['frac'],[ (k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']], d.get('latency_mode',{}).get('frames_per_s'))"; }
for spb in 1 2 4 8; do MST_A_SETS_PER_BLOCK=$spb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.5 2>/dev/null | show spb$spb; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.5 2>/dev/null | show default

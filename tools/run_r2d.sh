cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2d/pytest.log 2>&1
tail -4 gpurun_out/r2d/pytest.log
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1',d['value'],d['roofline']['frac'],[ (k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']], d.get('latency_mode',{}).get('frames_per_s'))"; }
MST_UPDATE_TIMING=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.5 2>gpurun_out/r2d/bench.err | show default
grep "tiles" gpurun_out/r2d/bench.err | head -3
bash tools/prof_stats.sh r2d --no-latency-leg --min-time 0.1 2>&1 | grep -E "^kernel|^k_(blend|warp_pyr|pyr_chain)"

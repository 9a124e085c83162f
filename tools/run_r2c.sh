cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2c/pytest.log 2>&1
tail -4 gpurun_out/r2c/pytest.log
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1',d['value'],d['roofline']['frac'],[ (k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']], d.get('latency_mode',{}).get('frames_per_s'))"; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.5 2>/dev/null | show default
bash tools/stamps.sh C3 8 2>&1 | grep -v amdgpu

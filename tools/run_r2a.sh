set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2a/pytest.log 2>&1
tail -5 gpurun_out/r2a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
cat gpurun_out/r2a/bench.json
for b in 1 2 3 4 6; do python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline --min-time 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch',$b,d['value'],d['roofline']['frac'],[ (k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']])"; done
python bench.py --steps 20 --warmup 5 --overlap --no-cpu-baseline --min-time 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap',d['value'],d['roofline']['frac'])"
python tools/time_update.py
nproc

# Round-end check on the GPU box (via gpurun): the -m gpu suite, the rocprofv3 passes behind profiles/
# (tools/profile_gpu.sh; its traffic.json is put in place first so that the bench lines of the same box carry it),
# bench.py (default, --batch 1, --undistort) and mst_update timing.  Outputs under gpurun_out/round and gpurun_out/prof_r02;
# copy what should be judged into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round
(time python -m pytest tests -m gpu -x -q) > gpurun_out/round/pytest.log 2>&1
tail -3 gpurun_out/round/pytest.log
bash tools/profile_gpu.sh r02 > gpurun_out/round/profile.log 2>&1
tail -3 gpurun_out/round/profile.log
bash tools/profile_gpu.sh r02b1 quick --batch 1 > gpurun_out/round/profile_b1.log 2>&1
cp gpurun_out/prof_r02/traffic.json profiles/traffic.json
python bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err
python -c "
import json; d=json.load(open('gpurun_out/round/bench.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], [(k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']], d.get('latency_mode'), d.get('overlapped'), d.get('host_buffers'), d.get('cpu_baseline'))"
python bench.py --batch 1 --no-cpu-baseline > gpurun_out/round/bench_batch1.json 2>/dev/null
python bench.py --undistort --no-cpu-baseline --no-latency-leg > gpurun_out/round/bench_undistort.json 2>/dev/null
python -c "
import json
for f in ('bench_batch1','bench_undistort'):
    d=json.load(open('gpurun_out/round/%s.json'%f)); print(f, d['value'], [(k['name'],k['ms_per_frame']) for k in d['roofline']['kernels']])"
python tools/time_update.py 2>&1 | tail -4

#!/bin/bash
# the N > 1 code path of bench.py on a one-GPU box: two ranks share the device, reductions over gloo (a smoke test, not a measurement)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export MST_BENCH_SMOKE_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --min-time 0.2 --no-camera-leg > gpurun_out/r05/bench_two_ranks_smoke.json 2> gpurun_out/r05/bench_two_ranks_smoke.err
echo "rc=$?"
tail -3 gpurun_out/r05/bench_two_ranks_smoke.err
python -c "
import json; d=json.loads(open('gpurun_out/r05/bench_two_ranks_smoke.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['scaling'], d['config']['parallelism'][:70], d.get('replicas'), d.get('ranks_seen',{}).get('world_size'), d.get('smoke'))"

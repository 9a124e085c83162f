#!/bin/bash
# Copy the round's judged artefacts from gpurun_out/ (scratch) into profiles/ (tracked): tools/collect_profiles.sh r03
TAG=${1:-r06}
for w in c3 c3_b1 C2 C5 C3-LUT C5-LUT; do
  d=gpurun_out/prof_${TAG}_$w
  [ -f $d/summary.txt ] || continue
  n=$(echo $w | tr 'A-Z-' 'a-z_')
  cp $d/summary.txt profiles/${TAG}_${n}_rocprofv3_summary.txt
  f=$(find $d/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f profiles/${TAG}_${n}_kernel_stats.csv
done
cp gpurun_out/round/bench.json profiles/${TAG}_c3_bench.json
cp gpurun_out/round/bench_batch1.json profiles/${TAG}_c3_bench_batch1.json
cp gpurun_out/round/bench_undistort.json profiles/${TAG}_c3_bench_undistort.json
cp gpurun_out/round/strip_model.jsonl profiles/${TAG}_strip_model.jsonl
cp gpurun_out/round/time_update.txt profiles/${TAG}_time_update.txt
for f in lone_call.json frame_loop.txt multi_gpu_loop.txt bench_strip_headline.json plan_stats.txt ingest_bench.json; do [ -f gpurun_out/round/$f ] && cp gpurun_out/round/$f profiles/${TAG}_$f; done
cp gpurun_out/traffic.json profiles/traffic.json
ls -la profiles/ | grep ${TAG}

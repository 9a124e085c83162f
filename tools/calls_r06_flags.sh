cd $GRAFT_REPO_ROOT
export VARIANT_LATENCY=
bash tools/variants.sh "base:" "maxilp:-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" "sched_ilp:-mllvm -amdgpu-sched-strategy=max-ilp" "sched_mem:-mllvm -amdgpu-sched-strategy=max-memory-clause" "base2:" 2>&1 | tee gpurun_out/r6_flags.txt

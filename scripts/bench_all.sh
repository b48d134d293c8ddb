#!/bin/bash
# one bench line per BASELINE.json config that fits one GPU (profiles/r04_bench_<name>.json);
# the headline config is the default command itself: python3 bench.py > profiles/r04_bench_full.json
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
hostname > gpurun_out/bench_all_host.txt; /opt/rocm/bin/rocm-smi --showserial 2>/dev/null | grep -i serial >> gpurun_out/bench_all_host.txt
for cfg in "nodcn --workload nodcn" "att_bs32 --workload att --batch 32" "resnet34 --workload resnet34" "resnet34_nodcn --workload resnet34_nodcn" "full_split --workload full --dtype split --steps 30" "full_f32 --workload full --dtype f32 --steps 10"; do
  set -- $cfg; name=$1; shift
  timeout -k 10 400 python bench.py "$@" --no-h2d > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err || echo "bench $name failed" | tee -a gpurun_out/bench_all_host.txt
  echo "done $name"
done

#!/bin/bash
# round-2 race bisection, call 1: reproduce + describe
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for m in forcesplit v1nosplit; do
  echo "=== MODE=$m default kernels" | tee -a gpurun_out/race1.log
  MODE=$m REPS=60 timeout -k 10 300 python scripts/race_probe.py >> gpurun_out/race1.log 2>&1 || echo "exit $?" >> gpurun_out/race1.log
done
echo "=== MODE=forcesplit GP_CONV_WINDOW=0" | tee -a gpurun_out/race1.log
GP_CONV_WINDOW=0 MODE=forcesplit REPS=60 timeout -k 10 300 python scripts/race_probe.py >> gpurun_out/race1.log 2>&1 || echo "exit $?" >> gpurun_out/race1.log
echo "=== MODE=forcesplit GP_GEMM_PP=0" | tee -a gpurun_out/race1.log
GP_GEMM_PP=0 MODE=forcesplit REPS=60 timeout -k 10 300 python scripts/race_probe.py >> gpurun_out/race1.log 2>&1 || echo "exit $?" >> gpurun_out/race1.log
echo "=== MODE='' (guarded default)" | tee -a gpurun_out/race1.log
MODE= REPS=60 timeout -k 10 300 python scripts/race_probe.py >> gpurun_out/race1.log 2>&1 || echo "exit $?" >> gpurun_out/race1.log
tail -5 gpurun_out/race1.log

#!/bin/bash
# round-2 race bisection, call 2: scratch-free build vs the round-1 build, same box
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/race2.log
: > $L
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -k "window or mlp or variant or pingpong or groupnorm" >> $L 2>&1 || { echo "pytest failed" >> $L; tail -30 $L; exit 1; }
R1=$PWD/givepose_amd/csrc/build/libgivepose_hip_r1.so
for m in v1nosplit forcesplit; do
  echo "=== NEW lib MODE=$m" >> $L
  MODE=$m REPS=300 EVENTS=3 timeout -k 10 300 python scripts/race_probe.py >> $L 2>&1 || echo "exit $?" >> $L
  echo "=== R1 lib MODE=$m" >> $L
  GP_LIB_PATH=$R1 MODE=$m REPS=300 EVENTS=3 timeout -k 10 300 python scripts/race_probe.py >> $L 2>&1 || echo "exit $?" >> $L
done
grep -E "===|buffers that ever|passed|failed" $L

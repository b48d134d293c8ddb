#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/race7.log
: > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { echo "pytest failed" >> $L; tail -40 $L; exit 1; }
tail -3 $L
for m in "" nosplit; do
  echo "=== default build MODE=$m" >> $L
  MODE=$m REPS=350 EVENTS=3 timeout -k 10 400 python scripts/race_probe.py >> $L 2>&1 || echo "exit $?" >> $L
done
grep -E "===|buffers that ever|^---" $L

#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/race4.log
: > $L
for cfg in "v1 k3" "v1dbg11 k3" "v1dbg12 k3" "v1dbg13 k3" "v1dbg14 k3"; do
  set -- $cfg
  AGG=$1 VIC=$2 NV=20 NA=12 ROUNDS=100 timeout -k 10 240 python scripts/race_min.py >> $L 2>&1 || echo "exit $? ($cfg)" >> $L
done
grep -v amdgpu.ids $L | grep AGG

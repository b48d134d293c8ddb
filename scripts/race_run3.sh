#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/race3.log
: > $L
for cfg in "v1 k3" "v7 k3" "v1small k3" "torchmm k3" "none k3" "v1 torchfma" "v1 gn"; do
  set -- $cfg
  AGG=$1 VIC=$2 NV=20 NA=12 ROUNDS=200 timeout -k 10 240 python scripts/race_min.py >> $L 2>&1 || echo "exit $? ($cfg)" >> $L
done
grep -v amdgpu.ids $L

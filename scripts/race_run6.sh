#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/race6.log
: > $L
for cfg in "v1dbg13 k3 0" "v1dbg13 k3 1" "v1 k3 0" "v1 k3 1"; do
  set -- $cfg
  GP_K3_NOPK=$3 AGG=$1 VIC=$2 NV=20 NA=12 ROUNDS=100 timeout -k 10 240 python scripts/race_min.py >> $L 2>&1 || echo "exit $? ($cfg)" >> $L
done
grep -v amdgpu.ids $L | grep -E "AGG|exit|Error"

#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/race5.log
: > $L
for cfg in "v1 k3" "v1dbg15 k3" "v10 k3" "v8 k3" "v7big k3" "v4 k3" "v13 k3" "mlp k3" "v1 gnapply" "v1dbg13 gnapply" "v10 gnapply" "mlp gnapply"; do
  set -- $cfg
  AGG=$1 VIC=$2 NV=20 NA=12 ROUNDS=100 timeout -k 10 240 python scripts/race_min.py >> $L 2>&1 || echo "exit $? ($cfg)" >> $L
done
grep -v amdgpu.ids $L | grep -E "AGG|exit|Error"

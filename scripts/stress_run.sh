#!/bin/bash
# overlapped-batches determinism stress of the DEFAULT build (split-K on), logs the GPU serial: profiles/r02_stress_<serial>.log
cd "$GRAFT_REPO_ROOT"
S=$(/opt/rocm/bin/rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | awk '{print $NF}')
L=gpurun_out/stress_$S.log
echo "GPU serial $S; $(date -u)" > $L
MODE= REPS=${REPS:-400} EVENTS=3 timeout -k 10 600 python scripts/race_probe.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L

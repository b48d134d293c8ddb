#!/bin/bash
# overlapped-batches determinism on this box: N runs of the bs-64 stress test (25 repetitions x 3 slots each) + race_probe
cd "$GRAFT_REPO_ROOT"
S=$(/opt/rocm/bin/rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | awk '{print $NF}')
n=0; N=${N:-16}
for i in $(seq 1 $N); do timeout -k 10 100 python -m pytest tests/test_hip_posenet.py -q -m gpu -k "bs64_stress" 2>&1 | tail -1 | grep -q failed && n=$((n+1)); done
echo "GPU serial $S: $n of $N runs of test_batches_in_flight_bs64_stress failed" | tee gpurun_out/stress_many_$S.log
REPS=${REPS:-300} bash scripts/stress_run.sh | tail -1 | cut -c1-200 | tee -a gpurun_out/stress_many_$S.log

#!/bin/bash
# sizing experiment: how the existing kernels scale with the batch per launch (round 3, grouped-launch planning)
set -e
mkdir -p gpurun_out
for b in 64 128 192 256; do
  for nf in 1 2; do
    python3 bench.py --group 1 --batch $b --inflight $nf --steps 40 --no-cpu-baseline --no-parity --no-h2d --kernels-out gpurun_out/bs_k_${b}_${nf}.json > gpurun_out/bs_${b}_${nf}.json 2> gpurun_out/bs_${b}_${nf}.err
    python3 -c "import json;d=json.load(open('gpurun_out/bs_${b}_${nf}.json'));print($b,$nf,d['value'],d['roofline']['frac'],d.get('one_batch_in_flight'))"
  done
done

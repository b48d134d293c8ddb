#!/bin/bash
# the XCD-contiguous patch order of the DCNv3 gather inside the real launch sequence (its input was just written by the projection GEMM)
for rep in 1 2; do for x in 1 0; do for g in 1 2; do
  GP_DCN_XCD=$x python3 bench.py --group $g --inflight 1 --steps 20 --no-cpu-baseline --no-parity --no-h2d --no-check --no-serial --kernels-out gpurun_out/dx_k.json > /dev/null 2>&1
  python3 -c "
import json
k=json.load(open('gpurun_out/dx_k.json'))
print('GP_DCN_XCD=$x group=$g rep=$rep', [(t['kernel'], t['avg_launch_us']) for t in k if t['kernel'].startswith('dcnv3')])"
done; done; done

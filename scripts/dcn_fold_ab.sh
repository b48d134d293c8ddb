#!/bin/bash
# Round 6: the fp16 DCNv3 gather with the mask weight folded into the corner weights (GP_DCN_FOLD=1) against the reference's association (default):
# alternating processes of scripts/dcn_bench.py at 64 crops.   Usage: scripts/dcn_fold_ab.sh [pairs]
N=${1:-3}
for i in $(seq 1 $N); do
  echo "pair $i arm A (default)"; python3 scripts/dcn_bench.py 2>/dev/null
  echo "pair $i arm B (GP_DCN_FOLD=1)"; GP_DCN_FOLD=1 python3 scripts/dcn_bench.py 2>/dev/null
done

#!/bin/bash
# Same-box A/B of the depth-wise 7x7 + LN kernels: GP_LIB_PATH=<baseline build> against the in-tree library, alternating, B = 64 / 128.
BASE=${1:-givepose_amd/libgivepose_hip_base.so}
for B in 64 128; do
  for rep in 1 2; do
    echo "--- baseline ($BASE) B=$B"; GP_LIB_PATH=$PWD/$BASE B=$B python3 scripts/dw_ab.py 2>&1 | grep "^C="
    echo "--- in-tree B=$B"; B=$B python3 scripts/dw_ab.py 2>&1 | grep "^C="
  done
done

#!/bin/bash
# Round 6: the fp16 DCNv3 gather with 16 bytes per lane (dcnv3_wave8_kernel, default) against the 8-bytes-per-lane kernel (GP_DCN_WAVE8=0):
# alternating processes of scripts/dcn_bench.py at 64 crops.   Usage: scripts/dcn_wave8_ab.sh [pairs]
N=${1:-3}
for i in $(seq 1 $N); do
  echo "pair $i arm A (GP_DCN_WAVE8=0: 8 bytes per lane)"; GP_DCN_WAVE8=0 python3 scripts/dcn_bench.py 2>/dev/null
  echo "pair $i arm B (default: 16 bytes per lane, DPP broadcasts)"; python3 scripts/dcn_bench.py 2>/dev/null
  echo "pair $i arm C (GP_DCN_LDSBC=1: 16 bytes per lane, LDS records)"; GP_DCN_LDSBC=1 python3 scripts/dcn_bench.py 2>/dev/null
done

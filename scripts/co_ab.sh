#!/bin/bash
# A/B on one box: which launches go to the 256x256 ping-pong tile when three batches are in flight (GP_GEMM_CO_MIN_TILES)
for rep in 1 2; do for t in 32 129 192 100000; do
  v=$(GP_GEMM_CO_MIN_TILES=$t python3 bench.py --group 1 --inflight 3 --steps 150 --no-cpu-baseline --no-parity --no-roofline --no-h2d 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['value'])")
  echo "min_tiles=$t rep=$rep value=$v"
done; done

#!/bin/bash
for rep in 1 2; do for cfg in "2 2" "2 3" "1 4" "4 1" "4 2"; do set -- $cfg
  v=$(python3 bench.py --group $1 --inflight $2 --steps 240 --no-cpu-baseline --no-parity --no-h2d --no-roofline --no-serial 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['overlap_check']['poses_bitwise_equal_to_serial_replay'])")
  echo "group $1 inflight $2 rep $rep: $v"
done; done

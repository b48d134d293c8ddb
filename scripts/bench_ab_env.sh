#!/bin/bash
# Same-box A/B of the whole forward with an environment switch: alternating processes, `value` of the short bench line.
# Usage: scripts/bench_ab_env.sh "GP_DW_TALL_MIN=1000000000" [pairs]   (arm A = the switch set, arm B = the tree's default)
SW="$1"; N=${2:-3}
for i in $(seq 1 $N); do
  for arm in A B; do
    if [ $arm = A ]; then v=$(env $SW python3 bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-f64 --no-roofline --no-parity --no-h2d --no-serial 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])");
    else v=$(python3 bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-f64 --no-roofline --no-parity --no-h2d --no-serial 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"); fi
    echo "pair $i arm $arm ($([ $arm = A ] && echo "$SW" || echo default)): $v images/s"
  done
done

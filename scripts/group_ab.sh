#!/bin/bash
# grouped launches end to end on one box: batches per launch x launches in flight
for cfg in "1 3" "2 2" "2 1" "3 1" "3 2" "1 3" "2 2"; do set -- $cfg
  python3 bench.py --group $1 --inflight $2 --steps 120 --no-cpu-baseline --no-parity --no-h2d > gpurun_out/grp_$1_$2.json 2> gpurun_out/grp_$1_$2.err || { echo "group $1 inflight $2 FAILED"; tail -3 gpurun_out/grp_$1_$2.err; }
  python3 -c "import json; d=json.load(open('gpurun_out/grp_$1_$2.json')); print('group $1 inflight $2', d['value'], d.get('one_launch_in_flight',{}).get('value'), d['one_batch_in_flight']['value'] if 'one_batch_in_flight' in d else None, d['roofline']['frac'], d['overlap_check'])"
done
python3 bench.py --group 2 --inflight 2 --steps 41 --no-cpu-baseline --no-parity --no-h2d --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('odd K', d['value'], d['steps'], d['overlap_check'])"

#!/bin/bash
# debug: bench.py's two-rank rehearsal (both ranks on cuda:0, gloo) with and without the small-M kernel; a ticker keeps the watchdog quiet
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04final
(while true; do sleep 45; echo tick; done) & T=$!
A="--gpus 2 --steps 6 --warmup 2 --batch 8 --inflight 2 --no-cpu-baseline --no-roofline --no-parity"
for e in ""; do
  env $e GP_BENCH_REHEARSE=1 timeout -k 10 150 python -u bench.py $A > gpurun_out/r04final/spawn_$$.json 2> gpurun_out/r04final/spawn_$$.err; echo "[$e] rc=$?"; grep -v amdgpu gpurun_out/r04final/spawn_$$.err | tail -4 | cut -c1-1200
done
kill $T

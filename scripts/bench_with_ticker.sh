#!/bin/bash
# python bench.py "$@" with a ticker on stdout beside it (keeps the pool's silence watchdog quiet during the CPU-baseline leg):
# gpurun -- 'bash scripts/bench_with_ticker.sh <out.json> [bench flags]'
cd "$GRAFT_REPO_ROOT"; out=$1; shift; mkdir -p "$(dirname "$out")"
(while true; do sleep 60; echo tick; done) & T=$!
timeout -k 10 400 python bench.py "$@" > "$out" 2> "${out%.json}.err"; rc=$?
kill $T; echo "bench rc=$rc"; tail -2 "${out%.json}.err"; exit $rc

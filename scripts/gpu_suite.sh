#!/bin/bash
# the GPU test suite with a ticker beside it (the suite prints a dot per test, but the first test alone runs a two-rank bench for a minute or two
# and the pool's watchdog kills a run that writes nothing for 7 minutes): gpurun -- 'bash scripts/gpu_suite.sh'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/suite
(while true; do sleep 60; date +%T >> gpurun_out/suite/tick.txt; done) & T=$!
PYTHONUNBUFFERED=1 timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu > gpurun_out/suite/pytest_gpu.txt 2>&1; rc=$?
kill $T; tail -4 gpurun_out/suite/pytest_gpu.txt; exit $rc

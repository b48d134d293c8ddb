#!/bin/bash
# Round-2 profiles of the benched code: kernel trace + stats of the bench command, PMC traffic passes, MFMA-busy pass.
# Everything lands under gpurun_out/r02_prof/; scripts/pmc_traffic.py / trace_summary.py / mfma_busy.py summarise.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
COMMIT=${1:-unknown}
O=gpurun_out/r02_prof
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
A="--no-cpu-baseline --no-parity --no-h2d"
echo "[1] kernel trace + stats of the bench command" | tee $O/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 10 --warmup 6 $A > $O/bench_kt.json 2> $O/bench_kt.err || { echo "kt failed" | tee -a $O/log.txt; tail -5 $O/bench_kt.err; exit 1; }
python3 scripts/trace_summary.py $O/kt > $O/step_kernels.txt 2>&1 || true
echo "[2] PMC FETCH_SIZE" | tee -a $O/log.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --inflight 1 $A > $O/bench_f.json 2> $O/bench_f.err || { echo "fetch failed" | tee -a $O/log.txt; tail -5 $O/bench_f.err; exit 1; }
echo "[3] PMC WRITE_SIZE" | tee -a $O/log.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --inflight 1 $A > $O/bench_w.json 2> $O/bench_w.err || { echo "write failed" | tee -a $O/log.txt; exit 1; }
python3 scripts/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json 4 $COMMIT > $O/pmc_traffic.txt 2>&1 || true
echo "[4] PMC MFMA busy" | tee -a $O/log.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --inflight 1 $A > $O/bench_s.json 2> $O/bench_s.err || { echo "sq failed" | tee -a $O/log.txt; exit 1; }
python3 scripts/mfma_busy.py $O/sq > $O/mfma_busy.txt 2>&1 || true
# keep only the summaries small enough to merge back
find $O -name "*.db" -delete; find $O -name "*.csv" -size +12M -delete; du -sh $O
for f in bench_kt bench_f; do echo "--- $f.err"; tail -6 $O/$f.err; done; ls -la $O | tee -a $O/log.txt
head -30 $O/mfma_busy.txt

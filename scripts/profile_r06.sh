#!/bin/bash
# Round-6 profiles of the benched code (run on the GPU box: gpurun -- 'bash scripts/profile_r06.sh <commit>').
# Everything lands under gpurun_out/r06_prof/; the summaries are copied into profiles/r06_* by hand (profiles/README.md).
#  [1] kernel trace + stats, strictly serial eager pass of the single-batch net (--group 1 --inflight 1 --no-graph)
#  [1b] the same of the launch sequence `value` runs (--group 2: 128 crops per launch): per-kernel averages comparable with bench.py's labels
#  [2] kernel trace + stats of the bench default (2 launch sequences of 2 batches in flight, hipGraph): what `value` runs
#  [3] kernel trace + stats of the split-operand parity mode (--dtype split, serial eager)
#  [4,5] PMC FETCH_SIZE / WRITE_SIZE passes (separate runs, as the microarchitecture guide prescribes)  [6] FETCH_SIZE calibration
#  [7] PMC MFMA-busy pass
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export GP_COMMIT=${1:-unknown}
O=gpurun_out/r06_prof
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
A="--no-cpu-baseline --no-parity --no-h2d"
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a $O/log.txt; }
step "[1] kernel trace, serial eager"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python3 bench.py --steps 10 --warmup 3 --group 1 --inflight 1 --no-graph --no-roofline $A > $O/bench_kt_serial.json 2> $O/bench_kt_serial.err || { step "kt_serial failed"; tail -5 $O/bench_kt_serial.err; exit 1; }
python3 scripts/trace_summary.py $O/kt_serial > $O/step_kernels_serial.txt 2>&1 || true
step "[1b] kernel trace, serial eager, grouped launch sequence (2 batches per launch)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_group -- python3 bench.py --steps 10 --warmup 4 --group 2 --inflight 1 --no-graph --no-roofline $A > $O/bench_kt_group.json 2> $O/bench_kt_group.err || { step "kt_group failed"; tail -5 $O/bench_kt_group.err; exit 1; }
python3 scripts/trace_summary.py $O/kt_group > $O/step_kernels_group.txt 2>&1 || true
step "[2] kernel trace, bench default (2 x 2 batches in flight, hipGraph)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_inflight -- python3 bench.py --steps 30 --warmup 6 --no-roofline $A > $O/bench_kt_inflight.json 2> $O/bench_kt_inflight.err || { step "kt_inflight failed"; tail -5 $O/bench_kt_inflight.err; exit 1; }
step "[3] kernel trace, split-operand mode (serial eager)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_split -- python3 bench.py --dtype split --steps 6 --warmup 3 --group 1 --inflight 1 --no-graph --no-roofline $A > $O/bench_kt_split.json 2> $O/bench_kt_split.err || { step "kt_split failed"; tail -5 $O/bench_kt_split.err; exit 1; }
python3 scripts/trace_summary.py $O/kt_split > $O/step_kernels_split.txt 2>&1 || true
P="--steps 4 --warmup 1 --no-graph --no-roofline --group 2 --inflight 1 --no-check --no-serial $A"     # 4 launches of 2 batches = 8 batches profiled
step "[4] PMC FETCH_SIZE (the benched launch sequence: 2 batches per launch)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py $P > $O/bench_f.json 2> $O/bench_f.err || { step "fetch failed"; tail -5 $O/bench_f.err; exit 1; }
step "[5] PMC WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py $P > $O/bench_w.json 2> $O/bench_w.err || { step "write failed"; exit 1; }
step "[6] FETCH_SIZE calibration (16 B / lane stream, 8 B / lane stream, 128-byte rows by 16 lanes x 8 B)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/pmc_calib.hip -o /tmp/pmc_calib && \
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -- /tmp/pmc_calib > $O/calib.out 2>&1 && python3 scripts/pmc_calib.py $O/calib > $O/pmc_calib.json 2>&1 || step "calibration failed"
python3 scripts/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json 8 $GP_COMMIT $O/pmc_calib.json 2 > $O/pmc_traffic.txt 2>&1 || true
step "[7] PMC MFMA busy"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 bench.py $P > $O/bench_s.json 2> $O/bench_s.err || { step "sq failed"; exit 1; }
python3 scripts/mfma_busy.py $O/sq > $O/mfma_busy.txt 2>&1 || true
# keep only what is small enough to merge back
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +12M -delete; find $O -name "*counter_collection.csv" -size +12M -delete; du -sh $O | tee -a $O/log.txt
for f in kt_serial kt_group kt_inflight kt_split; do ls $O/$f/*/ 2>/dev/null | head -5; done
head -12 $O/mfma_busy.txt; cat $O/pmc_calib.json

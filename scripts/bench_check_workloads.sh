#!/bin/bash
# bench.py's grouped-vs-separate check on every workload (quick legs only): gpurun -- 'bash scripts/bench_check_workloads.sh'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b2
for w in full resnet34 resnet34_nodcn nodcn; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-parity --no-h2d --no-roofline > gpurun_out/r04b2/$w.json 2> gpurun_out/r04b2/$w.err; echo "$w rc=$?"
  python3 - "$w" <<'PY'
import json, sys
l = json.loads(open(f"gpurun_out/r04b2/{sys.argv[1]}.json").read().strip().splitlines()[-1])
g = l["overlap_check"]["grouped_vs_separate_batches"]
print(l["value"], {k: (round(v, 5) if isinstance(v, float) else v) for k, v in g.items() if k != "bounds"})
PY
done

#!/bin/bash
# DCNv3 gather: XCD-contiguous workgroup order on / off -- time (scripts/dcn_bench.py) and HBM traffic (PMC FETCH_SIZE pass)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/dcn_xcd; rm -rf $O; mkdir -p $O
for x in 1 0; do
  echo "GP_DCN_XCD=$x"; GP_DCN_XCD=$x python3 scripts/dcn_bench.py 2>&1 | grep -v amdgpu
  GP_DCN_XCD=$x rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$x -- python3 scripts/dcn_bench.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/f$x/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "dcnv3" in r["Kernel_Name"]:
            acc[r["Grid_Size"]].append(float(r["Counter_Value"]) * 1024 * 2)
for g, v in sorted(acc.items(), key=lambda kv: -int(kv[0])):
    print(f"   grid {g}: HBM read bytes per launch (FETCH_SIZE x 2) {sum(v) / len(v) / 1e6:.1f} MB over {len(v)} launches")
PY
done
echo "algorithmic read bytes per launch: R=64: 134.2 (input) + 28.3 (offset / mask rows, fp32) = 162.5 MB; R=32: 40.6 MB; R=16: 10.2 MB"

#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... pass:
busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM-side kernel duration in shader cycles x SIMDs).  The counter is summed over the
chip's 1024 SIMDs and counts cycles (MI355X_MICROARCH.md, cycle-constants table); the duration comes from the dispatch's
own timestamps at the measured shader clock passed on the command line (default 2.4 GHz).
  python scripts/mfma_busy.py <dir> [GHz]"""
import collections, csv, glob, re, sys
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", name)[:64] + " grid=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            acc[k]["_n"] += 1
            acc[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for k, c in acc.items():
    if c["_n"] and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        cyc = c["_ns"] * ghz                       # shader cycles of all launches of this kernel (PMC runs serialise kernels)
        rows.append((c["_ns"], k, c["_n"], c["_ns"] / c["_n"] / 1e3, 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
for ns, k, n, us, busy in sorted(rows, reverse=True)[:24]:
    print(f"{k:96s} n={int(n):4d} avg={us:8.1f} us (profiled)  MFMA pipe busy {busy:5.1f} %")

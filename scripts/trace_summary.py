#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launches/avg/total of the LAST launch sequence of the run (from one
stem kernel to the next: one batch, or the G batches of a grouped launch)."""
import collections, csv, glob, sys
path = sys.argv[1]
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "stem_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]]   # stem_kernel / stem_mfma_kernel / resnet_stem_kernel: first launch of a step
assert len(stems) >= 2, "no two consecutive steps in the trace (no stem kernel found)"
step = rows[stems[-2]:stems[-1]]
agg = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"]
    n = n.replace("_ZN12_GLOBAL__N_1", "").split("(")[0][:48]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    key = (n, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])
    agg.setdefault(key, []).append(d)
tot = 0
for k, v in agg.items():
    tot += sum(v)
    print(f"{k[0]:48s} grid={int(k[1]) // int(k[2]):>6d}x{k[2]:>3s} vgpr={k[3]:>3s} lds={k[4]:>6s} n={len(v):3d} avg={sum(v) / len(v):8.1f} us tot={sum(v):9.1f}")
print("sum of kernel durations in one step: %.1f us; wall of step: %.1f us" % (tot, (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1000))

#!/usr/bin/env python3
"""FETCH_SIZE (KB) reported per launch of scripts/pmc_calib.hip's kernels over the bytes they really read -> the factor to
multiply FETCH_SIZE with for that access shape.   python scripts/pmc_calib.py <rocprofv3 -d dir>"""
import collections, csv, glob, json, sys
known = 1 << 30
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024)
res = {k: {"launches": len(v), "fetch_size_bytes": sum(v) / len(v), "known_bytes": known, "factor_to_apply": known / (sum(v) / len(v))} for k, v in acc.items()}
print(json.dumps(res, indent=1))

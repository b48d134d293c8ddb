"""Average PMC counter values per kernel from rocprofv3 --pmc output (counter_collection.csv files under a dir)."""
import csv, sys, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70] + " g" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if pat and pat not in k: continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")

#!/usr/bin/env python3
"""Timeline check of the pipelined H->D path: from a rocprofv3 --kernel-trace --memory-copy-trace run of `bench.py --h2d crops`,
how long the host->device copies take, how much of that time kernels are running beside them, and the gaps between
consecutive copies.   python scripts/h2d_trace.py <rocprofv3 -d dir>"""
import csv, glob, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
print("copy rows:", len(mc), "columns:", list(mc[0].keys()))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kt)
t_end = ks[-1][1]
t0 = t_end - (t_end - ks[0][0]) // 3          # last third of the run: steady state
kinds = {}
for r in mc:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    k = r.get("Direction", r.get("Name", "?"))
    kinds.setdefault(k, []).append((s, e, int(r.get("Size", 0) or 0)))
for k, v in kinds.items():
    v.sort()
    tot = sum(e - s for s, e, _ in v)
    by = sum(b for _, _, b in v)
    big = [(s, e, b) for s, e, b in v if b > 1 << 20]
    print(f"{k}: {len(v)} copies, {by / 1e6:.1f} MB, busy {tot / 1e6:.2f} ms of {(t_end - t0) / 1e6:.2f} ms window; "
          f"large copies: {len(big)}, avg {sum(e - s for s, e, _ in big) / max(1, len(big)) / 1e3:.1f} us, "
          f"avg rate {sum(b for _, _, b in big) / max(1, sum(e - s for s, e, _ in big)):.2f} GB/s")
busy = 0
cur_s, cur_e = None, None
for s, e in ks:
    if e < t0:
        continue
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels: some kernel running {100.0 * busy / (t_end - t0):.1f} % of the window")

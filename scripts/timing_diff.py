#!/usr/bin/env python3
"""Per-launch A/B of two GP_TIMING_DUMP files of the same eager pass (3 steps): python timing_diff.py a.txt b.txt [min_us]"""
import sys
from collections import defaultdict


def load(p):
    rows = [l.split() for l in open(p)]
    n = len(rows) // 3
    acc = defaultdict(list)
    for i, r in enumerate(rows):
        acc[i % n].append(float(r[3]))
    return [(rows[i][2], float(rows[i][4]), float(rows[i][5]), sorted(acc[i])[1]) for i in range(n)]


a, b = load(sys.argv[1]), load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
assert len(a) == len(b), (len(a), len(b))
ta = tb = 0.0
groups = defaultdict(lambda: [0, 0.0, 0.0])
for (na, fa, ba, ua), (nb, fb, bb, ub) in zip(a, b):
    ta += ua
    tb += ub
    g = groups[(na, fa, ba)]
    g[0] += 1
    g[1] += ua
    g[2] += ub
print(f"{'entry':22s} {'GFLOP':>9s} {'MB':>8s} {'n':>3s} {'A us':>9s} {'B us':>9s} {'A TF':>6s} {'B TF':>6s}  delta(total us)")
for (n, f, by), (c, ua, ub) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    if ua < thr and ub < thr:
        continue
    print(f"{n:22s} {f / 1e9:9.2f} {by / 1e6:8.1f} {c:3d} {ua / c:9.1f} {ub / c:9.1f} {f * c / ua / 1e6:6.0f} {f * c / ub / 1e6:6.0f}  {ub - ua:+9.1f}")
print(f"total A {ta:.1f} us   B {tb:.1f} us")

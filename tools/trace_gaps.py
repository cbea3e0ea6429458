#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv): per-kernel totals and the idle time between
consecutive kernels.  usage: trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]          # steady state only
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = gaps[(n0, n1)]
    g[0] += 1
    g[1] += max(0, s1 - e0)
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {100*(span-busy)/span:.1f}%")
per = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += 1
    per[n][1] += e - s
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {t/1e6:8.2f} ms  {c:5d} x {t/c/1e3:8.1f} us  {n}")
print("largest gaps (after -> before):")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {t/1e6:8.2f} ms  {c:5d} x {t/c/1e3:6.1f} us  {a[:28]} -> {b[:28]}")

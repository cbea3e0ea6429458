#!/usr/bin/env python3
"""Kernels of the LAST `ms` milliseconds of a rocprofv3 --kernel-trace csv run, grouped by name (steady state of a
loop whose first iterations tune or warm up): last_window.py <dir> <ms> [n]"""
import csv, glob, sys
from collections import defaultdict
d, ms, n = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
end = max(r[1] for r in rows)
win = [r for r in rows if r[0] >= end - ms * 1e6]
busy = sum(e - s for s, e, _ in win)
agg = defaultdict(lambda: [0, 0])
for s, e, k in win:
    agg[k][0] += e - s; agg[k][1] += 1
print(f"window {ms} ms: {len(win)} launches, busy {busy/1e6:.2f} ms")
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:n]:
    print(f"{t/1e6:8.3f} ms {t/busy*100:5.1f}% x{c:5d} avg {t/c/1e3:8.1f} us  {k[:120]}")

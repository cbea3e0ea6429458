#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv) of bench.py move by move: a move starts at each
obs_slot_write launch; for every move the span, the busy time, the idle share and the largest
kinds of gap between consecutive kernels, then per-kernel totals over the steady-state moves
(those without host-side pauses: idle below 5 %).   usage: trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
import re


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", n, 1)[0].split("::")[-1][:40]
starts = [i for i, r in enumerate(rows) if "obs_slot_write" in r[2]]
moves = [rows[a:b] for a, b in zip(starts, starts[1:]) if b - a > 20]
print(f"{len(rows)} kernels, {len(moves)} moves")
steady = []
for k, seg in enumerate(moves):
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
        g = gaps[(short(n0), short(n1))]
        g[0] += 1
        g[1] += max(0, s1 - e0)
    idle = 100 * (span - busy) / span
    top = sorted(gaps.items(), key=lambda kv: -kv[1][1])[:2]
    print(f"move {k}: {len(seg)} kernels, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {idle:.1f} %; largest gaps: "
          + "; ".join(f"{a} -> {b}: {c} x {t / c / 1e3:.1f} us" for (a, b), (c, t) in top))
    if idle < 5:
        steady.append(seg)
if steady:
    per = defaultdict(lambda: [0, 0])
    total = 0
    for seg in steady:
        total += seg[-1][1] - seg[0][0]
        for s, e, n in seg:
            per[short(n)][0] += 1
            per[short(n)][1] += e - s
    print(f"steady-state moves: {len(steady)}, {total / len(steady) / 1e6:.2f} ms per move; per move and kernel:")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"  {t / len(steady) / 1e6:7.3f} ms  {c // len(steady):4d} x {t / c / 1e3:8.1f} us  {n}")

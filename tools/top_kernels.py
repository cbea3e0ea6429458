#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: top_kernels.py <dir> [n]"""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {float(r["TotalDurationNs"])/tot*100:5.1f}% calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')

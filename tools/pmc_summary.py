#!/usr/bin/env python3
"""Mean per-dispatch value of every counter in a rocprofv3 --pmc output directory, for kernels whose
name contains a substring.   usage: pmc_summary.py <dir> <kernel substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
vals = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(path)):
    if sub in r["Kernel_Name"]:
        vals[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for name, per in sorted(vals.items()):
    v = list(per.values())
    print(f"{name:32s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")

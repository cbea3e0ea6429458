#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs of tools/profile_round.sh into the records kept under profiles/:
<tag>_kernel_stats.csv, <tag>_kernels.md (table + dynamics-launch timing) and traffic.json (HBM-side
bytes per dynamics launch of the dominant kernel).  They are written beside the raw data; copy them."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

out, tag = sys.argv[1], sys.argv[2]
prof = out


def one(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"no file matches {pattern} under {out}")
    return hits[0]


stats = one("stats/**/*kernel_stats.csv")
shutil.copy(stats, os.path.join(prof, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
bench = json.loads(open(os.path.join(out, "bench_stats.json")).read().strip().splitlines()[-1])

# dynamics launches of the chain kernel = the short ones (5 layers); the representation tail has 18
trace = list(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))))
chain = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if "conv_chain_" in r["Kernel_Name"]]
cut = 2 * statistics.median(chain)
dyn = [d for d in chain if d < cut]
tail = [d for d in chain if d >= cut]

lines = [f"# {tag}: kernel statistics", "",
         "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` "
         f"({bench['config']['workload']}); bench.py also runs 2 further moves to time the dominant kernel.", "",
         "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
for r in rows[:16]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    lines.append(f"| `{name[:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
lines += ["", f"chain kernel (`conv_chain_pair_kernel` with more images than CUs, else `conv_chain_kernel`), dynamics launches (5 layers): {len(dyn)} launches, mean {statistics.mean(dyn) / 1e3:.1f} us, "
          f"median {statistics.median(dyn) / 1e3:.1f} us; representation-tail launches (18 layers): {len(tail)}, "
          f"mean {statistics.mean(tail) / 1e3:.1f} us." if tail else "",
          f"bench.py's HIP-event timing of the dynamics launches in the same run: {bench['roofline']['launch_us']:.1f} us "
          f"({bench['roofline']['achieved']:.0f} TFLOP/s, frac {bench['roofline']['frac']:.3f}); "
          f"throughput under the profiler {bench['value']:.0f} sims/s."]


def pmc_mean(kind, counter):
    path = one(f"{kind}/**/*counter_collection.csv")
    vals = {}
    for r in csv.DictReader(open(path)):
        if "conv_chain_" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(vals.values())
    v = [x for x in v if x < 2 * statistics.median(v)]      # dynamics launches only
    return statistics.mean(v), len(v)


try:
    fetch_kib, n_f = pmc_mean("fetch", "FETCH_SIZE")
    write_kib, n_w = pmc_mean("write", "WRITE_SIZE")
    B = bench["config"]["games_per_gpu"]
    fetch_b, write_b = fetch_kib * 1024 * 2, write_kib * 1024     # gfx950: 128-B read requests are tallied at 64 B
    algo = B * 2 * 144 * 256 * 2 + 5 * 9 * 256 * 256 * 2 + B * 256 * 4
    traffic = {"kernel": "conv_chain_pair_kernel (dynamics, 5 layers)", "games_per_gpu": B,
               "hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
               "algorithmic_bytes_per_launch": algo, "launches_averaged": [n_f, n_w],
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1 "
                         "--warmup 1; counters are in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at "
                         "64 B); mean over the dynamics launches", "profile": tag}
    json.dump(traffic, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    lines += ["", f"HBM-side traffic per dynamics launch (PMC, separate passes): FETCH_SIZE x2 = {fetch_b / 1e6:.1f} MB, "
              f"WRITE_SIZE = {write_b / 1e6:.1f} MB; algorithmic = {algo / 1e6:.1f} MB."]
except SystemExit as e:
    lines += ["", f"PMC passes missing: {e}"]
open(os.path.join(prof, f"{tag}_kernels.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

#!/usr/bin/env python3
"""Per-move time budget of the benchmark's workload from a rocprofv3 kernel-statistics file of
`bench.py --steps 3 --warmup 1 --no-extras` (6 moves in all: 1 warm-up, 3 timed, 2 with HIP events around the
dominant kernel) plus the dynamics / tail split that tools/summarize_profile.py wrote.
usage: move_budget.py profiles/r03_kernel_stats.csv profiles/r03_kernels.md [moves]"""
import csv
import re
import sys

stats, notes = sys.argv[1], sys.argv[2]
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(stats))}
txt = open(notes).read()
m = re.search(r"dynamics launches \(5 layers\): (\d+) launches, mean ([\d.]+) us, median ([\d.]+) us; representation-tail launches "
              r"\(18 layers\): (\d+), mean ([\d.]+) us", txt)
n_dyn, dyn_mean, dyn_med, n_tail, tail_mean = int(m[1]), float(m[2]), float(m[3]), int(m[4]), float(m[5])
bench = re.search(r"throughput under the profiler (\d+) sims/s", txt)
sims_s = float(bench[1])


def find(sub, nth=0):
    hits = [(k, v) for k, v in rows.items() if sub in k]
    return hits[nth][1] if hits else (0, 0.0)


items = [("dynamics network: conv_chain_pair_kernel, 5 layers", n_dyn / moves, dyn_mean),
         ("representation tail: conv_chain_pair_kernel, 18 layers", n_tail / moves, tail_mean),
         ("first convolution on the rotating observation (conv3x3 kernel, 96x96x96 -> 48x48x256)",) + tuple(
             x / (moves if i == 0 else 1) for i, x in enumerate(find("CfgILi96ELi96ELi256"))),
         ("24x24x128 residual blocks: conv_mid_chain_kernel",) + tuple(
             x / (moves if i == 0 else 1) for i, x in enumerate(find("conv_mid_chain"))),
         ("48x48x256 -> 24x24x128 stride 2",) + tuple(x / (moves if i == 0 else 1) for i, x in enumerate(find("CfgILi256ELi64ELi128"))),
         ("24x24x128 -> 12x12x256 stride 2",) + tuple(x / (moves if i == 0 else 1) for i, x in enumerate(find("CfgILi128ELi128ELi256"))),
         ("heads_kernel (policy / value / reward heads, softmax, support decode)",) + tuple(
             x / (moves if i == 0 else 1) for i, x in enumerate(find("heads_kernel"))),
         ("tree step: expand_backup_select_staged_kernel",) + tuple(
             x / (moves if i == 0 else 1) for i, x in enumerate(find("expand_backup_select_staged"))),
         ("last backup of a move: expand_backup_kernel",) + tuple(x / (moves if i == 0 else 1) for i, x in enumerate(find("expand_backup_kernel"))),
         ("action choice: select_action_kernel",) + tuple(x / (moves if i == 0 else 1) for i, x in enumerate(find("select_action_kernel"))),
         ("new frame into the rotating observation: obs_slot_write_kernel", 1.0, find("obs_slot_write")[1]),
         ("new frame into the u8 ring: ring_push_kernel", 1.0, find("ring_push")[1])]
print("| per move (1024 games x 50 simulations) | launches | us each | ms |\n|---|---|---|---|")
total = 0.0
for name, n, us in items:
    ms = n * us / 1e3
    total += ms
    print(f"| {name} | {n:.0f} | {us:.1f} | {ms:.3f} |")
other = 0.15
print(f"| root_init / first select / read-out, noise and uniform copies, fills (~60 launches of 3-6 us) | | | ~{other:.2f} |")
print(f"| **sum of kernel time** | | | **{total + other:.2f}** |")
print(f"| measured in the same run (under the profiler): {sims_s:.0f} sims/s | | | **{1024 * 50 / sims_s * 1e3:.2f}** |")
print(f"\n(dynamics launches: mean {dyn_mean:.1f} us, median {dyn_med:.1f} us - the mean carries a few launches that straddle "
      f"clock steps; with the median the sum is {total + other - n_dyn / moves * (dyn_mean - dyn_med) / 1e3:.2f} ms.)")

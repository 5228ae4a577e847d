#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (our kernels in full, everything else summed)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ours, other = [], 0.0
for r in rows:
    name = r["Name"]
    if "w2::" in name:
        short = name.split("w2::", 1)[1].split("(")[0]
        ours.append((short, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
    else:
        other += float(r["TotalDurationNs"]) / 1e6
print("| kernel | calls | total ms | avg ms | % of all GPU time |")
print("|---|---|---|---|---|")
for s, c, t, a, p in sorted(ours, key=lambda x: -x[2]):
    print(f"| {s} | {c} | {t:.3f} | {a:.4f} | {p:.2f} |")
print(f"| (torch data generation, rocPRIM, memset/copy) | - | {other:.3f} | - | - |")

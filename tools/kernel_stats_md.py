#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats: <dir>/**/*_kernel_stats.csv -> markdown table + filtered CSV of our kernels.
usage: kernel_stats_md.py <rocprof out dir> <out.md> <out.csv> [rocprim]     (rocprim: list rocPRIM's sort / scan kernels too)"""
import csv, glob, sys
src = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with_rocprim = len(sys.argv) > 4 and sys.argv[4] == "rocprim"
ours = [r for r in rows if "w2::" in r["Name"] or r["Name"].startswith("k_") or (with_rocprim and "rocprim" in r["Name"])]
with open(sys.argv[3], "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in ours: w.writerow(r)
with open(sys.argv[2], "w") as f:
    f.write("| kernel | calls | total ms | avg ms | % of all GPU time |\n|---|---|---|---|---|\n")
    for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void w2::", "").replace("w2::", "").split("(")[0][:110]
        f.write(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {100*float(r['TotalDurationNs'])/tot:.2f} |\n")
    other = tot - sum(float(r["TotalDurationNs"]) for r in ours)
    f.write(f"| (torch data generation, rocPRIM, memset/copy) | - | {other/1e6:.3f} | - | - |\n")

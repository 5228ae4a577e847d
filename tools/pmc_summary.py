#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into a markdown table for our kernels.
usage: pmc_summary.py <dir> [<dir> ...]   (one directory per --pmc pass; kernel launches are summed)"""
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "w2::" not in name:
                continue
            short = name.split("w2::", 1)[1].split("(")[0]
            acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(short, path)].add(r["Dispatch_Id"])
cols = ["launches", "FETCH_SIZE GB (x2 corr.)", "WRITE_SIZE GB", "VALU inst", "SALU inst", "LDS inst", "LDS conflict %", "wait %", "active %"]
print("| kernel | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
order = sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))
for k in order:
    a = acc[k]
    if a.get("SQ_WAVE_CYCLES", 0) < 1e9:
        continue
    n = max(len(v) for (kk, p), v in disp.items() if kk == k)
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    row = [str(n), f"{a.get('FETCH_SIZE', 0) * 1024 / 1e9:.2f} ({a.get('FETCH_SIZE', 0) * 2048 / 1e9:.2f})", f"{a.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f}",
           f"{a.get('SQ_INSTS_VALU', 0):.3g}", f"{a.get('SQ_INSTS_SALU', 0):.3g}", f"{a.get('SQ_INSTS_LDS', 0):.3g}",
           f"{100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 0), 1):.0f}",
           f"{100 * a.get('SQ_WAIT_ANY', 0) / wc:.0f}", f"{100 * a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f}"]
    print(f"| {k} | " + " | ".join(row) + " |")

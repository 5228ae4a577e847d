#!/usr/bin/env python3
"""Timeline of the LAST run of a phase in a rocprofv3 kernel trace: every kernel between the last launch of <first_kernel> and the last
launch of <last_kernel> with its start (ms from the first), duration and the idle gap in front of it.
usage: kernel_timeline.py <trace dir> <first_kernel substring> <last_kernel substring> [min gap us to flag = 15]"""
import csv, glob, sys
d, first, last = sys.argv[1], sys.argv[2], sys.argv[3]
flag_us = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("w2::", "")
    n = n.split("(")[0]
    if "rocprim" in n: n = "rocprim::" + n.split("::")[-1].split("<")[0] + ("<" + n.split("<")[1][:40] if "<" in n else "")
    return n[:70]
i1 = max(i for i, r in enumerate(rows) if last in r[2])
i0 = max(i for i, r in enumerate(rows[:i1]) if first in r[2])
t0 = rows[i0][0]
busy = 0; gaps = 0; prev_end = rows[i0][0]
print(f"{'start ms':>9} {'dur us':>9} {'gap us':>8}  kernel")
for s, e, n in rows[i0:i1 + 1]:
    gap = (s - prev_end) / 1e3
    busy += (e - s); gaps += max(0, s - prev_end)
    print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f} {gap:8.1f}{' *' if gap > flag_us else '  '} {short(n)}")
    prev_end = max(prev_end, e)
print(f"wall {(rows[i1][1] - t0) / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle gaps {gaps / 1e6:.3f} ms over {i1 - i0 + 1} launches")

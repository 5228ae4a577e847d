#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into a markdown table for our kernels.
usage: pmc_summary.py <dir> [<dir> ...]   (one directory per --pmc pass; kernel launches are summed)

Columns: launches; total ms (kernel-trace timestamps of the first pass); FETCH_SIZE / WRITE_SIZE in GB (FETCH also x2, the
gfx950 correction of MI355X_MICROARCH.md for wide coalesced reads); wave-instructions by unit; per-wave percentages of
SQ_WAVE_CYCLES: waiting, any instruction active, VALU active; `issue` = active% x resident waves per SIMD (how busy the
SIMD's issue port is: ~100 % = instruction-issue-bound whatever each wave's own wait% says); LDS bank-conflict share of
LDS-active cycles."""
import csv, sys, glob, collections, json
json_out = None
if "--json" in sys.argv:                      # --json <file>: the same numbers as a JSON object (what bench.py reads as profiles/pmc_step2.json)
    i = sys.argv.index("--json"); json_out = sys.argv[i + 1]; del sys.argv[i:i + 2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float); seen = set(); first_pass = {}
first = sys.argv[1]
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "w2::" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").split("w2::", 1)[1].split("(")[0]
            # a counter collected in several passes (SQ_WAVE_CYCLES rides along in two of them) is taken from the first pass that has it
            src = first_pass.setdefault((short, r["Counter_Name"]), path)
            if src == path:
                acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(short, path)].add(r["Dispatch_Id"])
            if path == first and (path, r["Dispatch_Id"]) not in seen:
                seen.add((path, r["Dispatch_Id"]))
                dur[short] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                acc[short]["_waves"] += float(r["Grid_Size"]) / 64
                acc[short]["_wg"] = float(r["Workgroup_Size"])
cols = ["launches", "ms", "FETCH GB (x2)", "WRITE GB", "VALU", "SALU", "LDS", "VMEM", "wait %", "active %", "VALU act %", "LDS conflict %"]
print("| kernel | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
for k in sorted(acc, key=lambda k: -dur[k]):
    a = acc[k]
    if dur[k] < 0.5:
        continue
    n = max(len(v) for (kk, p), v in disp.items() if kk == k)
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    row = [str(n), f"{dur[k]:.2f}", f"{a.get('FETCH_SIZE', 0) * 1024 / 1e9:.2f} ({a.get('FETCH_SIZE', 0) * 2048 / 1e9:.2f})", f"{a.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f}",
           f"{a.get('SQ_INSTS_VALU', 0):.3g}", f"{a.get('SQ_INSTS_SALU', 0):.3g}", f"{a.get('SQ_INSTS_LDS', 0):.3g}", f"{a.get('SQ_INSTS_VMEM', 0):.3g}",
           f"{100 * a.get('SQ_WAIT_ANY', 0) / wc:.0f}", f"{100 * a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f}", f"{100 * a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.1f}",
           f"{100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 0), 1):.0f}"]
    print(f"| {k} | " + " | ".join(row) + " |")
if json_out:
    ks = {}
    for k in acc:
        a = acc[k]
        if dur[k] < 0.05:
            continue
        n = max(len(v) for (kk, p), v in disp.items() if kk == k)
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        ks[k] = {"launches": n, "ms": round(dur[k], 3), "fetch_bytes": a.get("FETCH_SIZE", 0) * 1024, "write_bytes": a.get("WRITE_SIZE", 0) * 1024,
                 "insts_valu": a.get("SQ_INSTS_VALU", 0), "insts_salu": a.get("SQ_INSTS_SALU", 0), "insts_lds": a.get("SQ_INSTS_LDS", 0), "insts_vmem": a.get("SQ_INSTS_VMEM", 0),
                 "wait_frac": a.get("SQ_WAIT_ANY", 0) / wc, "active_frac": a.get("SQ_ACTIVE_INST_ANY", 0) / wc, "valu_active_frac": a.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                 "waves": a.get("_waves", 0), "workgroup": a.get("_wg", 0)}
    json.dump({"source": "rocprofv3 --pmc passes (tools/r04_pmc.sh) over one whole Step 2, python3 tools/gpu_pmc_target.py 5e7 1 step2; summed per kernel over its launches of ONE step",
               "units": "bytes as FETCH_SIZE / WRITE_SIZE x 1024 (uncorrected: bench.py applies the gfx950 x2 to FETCH); instruction counts are wave-instructions; *_frac are of SQ_WAVE_CYCLES",
               "passes": sys.argv[1:], "kernels": ks}, open(json_out, "w"), indent=1, sort_keys=True)

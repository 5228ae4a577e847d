#!/bin/bash
# PMC passes over one whole Step 2 at 50 M reads (separate runs per counter group, kernel trace + stats in a run of their own):
#   tools/r04_pmc.sh <tag> [env assignments...]   -> gpurun_out/pmc_<tag>.md, gpurun_out/stats_<tag>.csv
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $out/sq -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step2 > $out/sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $out/lds -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step2 > $out/lds.log 2>&1
if [ -z "$NO_TRAFFIC" ]; then
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step2 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step2 > $out/write.log 2>&1
fi
python3 tools/pmc_summary.py --json gpurun_out/pmc_$tag.json $out/sq $out/lds $out/fetch $out/write > gpurun_out/pmc_$tag.md 2>$out/summary.err
python3 - $out/lds >> gpurun_out/pmc_$tag.md <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "w2::" in n:
            acc[n.split("w2::", 1)[1].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
print("\n| kernel | LDS inst active % of wave cycles | wait inst LDS % | wait inst any % | LDS idx active / busy cycles | bank conflict / idx active |\n|---|---|---|---|---|---|")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = a.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"| {k} | {100*a.get('SQ_ACTIVE_INST_LDS',0)/wc:.1f} | {100*a.get('SQ_WAIT_INST_LDS',0)/wc:.1f} | {100*a.get('SQ_WAIT_INST_ANY',0)/wc:.1f} | {a.get('SQ_LDS_IDX_ACTIVE',0)/max(a.get('SQ_BUSY_CYCLES',1),1):.3f} | {a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} |")
PY
rm -rf $out/*/*/*.db 2>/dev/null
du -sh $out | tail -1

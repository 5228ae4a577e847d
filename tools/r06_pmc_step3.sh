#!/bin/bash
# PMC passes over one Step 3 behind Step 2 at 50 M diploid reads (tools/r04_pmc.sh's recipe: a run per counter group, FETCH and WRITE alone)
export TMPDIR=/tmp
out=gpurun_out/pmc_step3_r06
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $out/sq -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step3 > $out/sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $out/lds -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step3 > $out/lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step3 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o x --output-format csv -- python3 tools/gpu_pmc_target.py 5e7 1 step3 > $out/write.log 2>&1
python3 tools/pmc_summary.py $out/sq $out/lds $out/fetch $out/write 2>$out/summary.err | grep -E "^\| kernel|^\|---|k3_|k_rank|k_scan|k_rs_|k_split" > gpurun_out/r06_pmc_step3.md
rm -rf $out/*/*/*.db 2>/dev/null
head -30 gpurun_out/r06_pmc_step3.md

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/$1 -- python3 $R/tools/gpu_pmc_target.py 5e7 1 > $R/gpurun_out/$1.log 2>&1; }
run pmc_a "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
run pmc_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
run pmc_c "FETCH_SIZE"
run pmc_d "WRITE_SIZE"
cd $R; for x in a b c d; do grep -c . gpurun_out/pmc_$x/*/*counter_collection.csv; done

bash tools/r04_budget.sh
( time python tools/gpu_parity_at_scale.py 5e7 ) > gpurun_out/r04_parity_50M_reads.txt 2>&1

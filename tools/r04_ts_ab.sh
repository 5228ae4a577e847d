#!/bin/bash
# k_rank_tiles with a conflict-free node layout, incremental reverse complements in the two prune kernels: parity, then bench
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/k1_parity.txt 2>&1; tail -3 gpurun_out/k1_parity.txt >> gpurun_out/k3ab.log
echo "== default" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh 20
echo "== default again" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh 20

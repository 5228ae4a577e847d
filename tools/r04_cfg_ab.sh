#!/bin/bash
# k_table_insert with four k-mers per thread: parity, then the counting-kernel shapes 20 / 22 / 24 (the last one had the insert kernel at half speed)
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/k1_parity.txt 2>&1; tail -3 gpurun_out/k1_parity.txt >> gpurun_out/k3ab.log
for rep in 1 2; do for c in 20 22 24; do echo "== cfg $c (rep $rep)" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh $c; done; done

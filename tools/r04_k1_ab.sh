#!/bin/bash
# the whole GPU suite on the new defaults (lane-per-read K1 that cuts only full records, 32-byte records), then the bench workload:
# default, ALIGN64 cuts, the wavefront-per-read kernel
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/k1_pytest_gpu.txt 2>&1; tail -5 gpurun_out/k1_pytest_gpu.txt >> gpurun_out/k3ab.log
echo "== default" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh 20
echo "== default, BATCHES=1" >> gpurun_out/k3ab.log; W2RAP_BATCHES=1 bash tools/r04_k3_ab.sh 20
echo "== ALIGN64 cuts" >> gpurun_out/k3ab.log; W2RAP_K1_ALIGN64=1 bash tools/r04_k3_ab.sh 20
echo "== wave kernel" >> gpurun_out/k3ab.log; W2RAP_K1=wave bash tools/r04_k3_ab.sh 20

#!/bin/bash
# K1 A/B on one box (profiles/r04_k1_lane_ab.txt): the whole GPU suite, then the bench workload with the lane-per-read kernel (default:
# records cut only when full), with the wavefront kernel's cuts (W2RAP_K1_ALIGN64), and with the wavefront-per-read kernel (W2RAP_K1=wave)
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/k1_pytest_gpu.txt 2>&1; tail -5 gpurun_out/k1_pytest_gpu.txt >> gpurun_out/k3ab.log
echo "== default" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh 20
echo "== default, BATCHES=1" >> gpurun_out/k3ab.log; W2RAP_BATCHES=1 bash tools/r04_k3_ab.sh 20
echo "== ALIGN64 cuts" >> gpurun_out/k3ab.log; W2RAP_K1_ALIGN64=1 bash tools/r04_k3_ab.sh 20
echo "== wave kernel" >> gpurun_out/k3ab.log; W2RAP_K1=wave bash tools/r04_k3_ab.sh 20

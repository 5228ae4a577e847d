#!/bin/bash
# schedule knobs after the partition got faster: last read batch shorter, number of bucket slices
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
echo "== default" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh 20
for f in 0.5 0.3; do echo "== LAST_BATCH=$f" >> gpurun_out/k3ab.log; W2RAP_LAST_BATCH=$f bash tools/r04_k3_ab.sh 20; done
for s in 3 6 8; do echo "== SLICES=$s" >> gpurun_out/k3ab.log; W2RAP_SLICES=$s bash tools/r04_k3_ab.sh 20; done
echo "== BATCHES=3" >> gpurun_out/k3ab.log; W2RAP_BATCHES=3 bash tools/r04_k3_ab.sh 20

#!/bin/bash
# k-mers per bucket sweep on the bench workload (the lane K1 cuts 17 % fewer records, so a bucket of the same k-mers has fewer of them)
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
for kpb in 4500 5200 6000 7000 3800; do
  echo "== KPB=$kpb" >> gpurun_out/k3ab.log; W2RAP_KPB=$kpb bash tools/r04_k3_ab.sh 20
done

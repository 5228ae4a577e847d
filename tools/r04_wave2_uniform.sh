#!/bin/bash
# the uniform bench workload with the wave-per-read pather for the listed reads (mode 2) at part budgets 12 / 6 / 4, against the default
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
echo "== default (listed lane kernel, budget 12)" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh 22
for b in 12 6 4; do echo "== PATH_WAVE=2 budget $b" >> gpurun_out/k3ab.log; W2RAP_PATH_WAVE=2 W2RAP_PATH_BUDGET=$b bash tools/r04_k3_ab.sh 22; done

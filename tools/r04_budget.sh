#!/bin/bash
# read-pathing part budget (parts a read may cut into before the first pass hands it to the listed pass): uniform and planted workload
rm -f gpurun_out/r04_budget.txt
for b in 12 10 8 6; do
  echo "== W2RAP_PATH_BUDGET=$b uniform" >> gpurun_out/r04_budget.txt
  W2RAP_PATH_BUDGET=$b timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); k=r['kernel_ms_per_step']
print('step', round(r['ms_per_step'],1), 'path phase', round(r['phase_ms']['path'],2), {x: round(v,2) for x,v in k.items() if 'path' in x})" >> gpurun_out/r04_budget.txt
  echo "== W2RAP_PATH_BUDGET=$b planted" >> gpurun_out/r04_budget.txt
  W2RAP_PATH_BUDGET=$b python tools/gpu_planted_check.py 5e7 2>/dev/null | grep "path ms\|kernels" >> gpurun_out/r04_budget.txt
done

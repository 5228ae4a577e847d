#!/bin/bash
# shape 22 (512-record tile) against the default 20 (640) on the PLANTED workload too (skewed buckets): the bench line with extras, two steps
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
for c in 20 22; do
  echo "== cfg $c" >> gpurun_out/k3ab.log
  W2RAP_K3=$c W2RAP_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/cfg_$c.err | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print(json.dumps({'ms':r['ms_per_step'],'phase':r['phase_ms'],'planted':r['planted_workload']['ms_per_step'],'planted_phase':r['planted_workload']['phase_ms'],'frac':r['roofline']['frac']}))" >> gpurun_out/k3ab.log 2>&1
  grep -h "deferred" gpurun_out/cfg_$c.err | sort | uniq -c | sort -rn | head -4 >> gpurun_out/k3ab.log
done

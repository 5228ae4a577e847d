#!/bin/bash
# A/B of the counting-kernel shapes on the bench workload (50 M reads): W2RAP_K3 = 0 (rounds 1-3), 20..23 (round 4)
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== W2RAP_K3=$cfg" >> gpurun_out/k3ab.log
  W2RAP_K3=$cfg W2RAP_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/k3ab_$cfg.err | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print(json.dumps({'ms':r['ms_per_step'],'phase':r['phase_ms'],'roof':{k:r['roofline'][k] for k in ('kernel','frac','avg_launch_ms','launches_per_step','count_phase_frac')},'alone':r['roofline'].get('not_overlapped'),'kern':r['kernel_ms_per_step'],'S':r['config']['kmers_solid'],'D':r['config']['kmers_distinct'],'paths':r['config']['path_elements']}))" >> gpurun_out/k3ab.log 2>&1
  grep -h "deferred\|count:" gpurun_out/k3ab_$cfg.err | tail -3 >> gpurun_out/k3ab.log
done

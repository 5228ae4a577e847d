#!/bin/bash
# the wave-per-read pather with the lanes on the parts in its second stage (W2RAP_PATH_WAVE=2): tests, then the planted workload against
# the oracle at 1 M and 2 M reads and timed at 50 M, with the listed lane kernel (default), mode 1 and mode 2; then the part budgets with mode 2
mkdir -p gpurun_out; rm -f gpurun_out/wave2.txt
timeout 1500 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "retry_and_fallback" > gpurun_out/wave2_tests.txt 2>&1; tail -3 gpurun_out/wave2_tests.txt >> gpurun_out/wave2.txt
for m in 0 1 2; do
  echo "== W2RAP_PATH_WAVE=$m" >> gpurun_out/wave2.txt
  sizes="5e7"; [ $m = 2 ] && sizes="1e6 2e6 5e7"
  W2RAP_PATH_WAVE=$m W2RAP_TRACE=1 timeout 1200 python3 tools/gpu_planted_check.py $sizes 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|count:\|list ranking\|drop_results\|deferred" >> gpurun_out/wave2.txt
done
for b in 6 4 3; do
  echo "== W2RAP_PATH_WAVE=2 W2RAP_PATH_BUDGET=$b" >> gpurun_out/wave2.txt
  W2RAP_PATH_WAVE=2 W2RAP_PATH_BUDGET=$b timeout 900 python3 tools/gpu_planted_check.py 5e7 2>&1 | grep "path ms\|kernels" >> gpurun_out/wave2.txt
done

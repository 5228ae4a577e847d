timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -x -q -k "internal_retry or replay or stages or tail or second_build or shapes" 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > gpurun_out/r04_parity5.log
W2RAP_PATH_WAVE=1 W2RAP_TRACE=1 python tools/gpu_planted_check.py 2e6 5e7 > gpurun_out/r04_planted_wave.txt 2>&1
rm -f gpurun_out/k3ab.log
tools/r04_k3_ab.sh 20

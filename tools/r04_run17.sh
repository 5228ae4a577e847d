timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_two_ranks.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -6 > gpurun_out/r04_k1win.log
rm -f gpurun_out/k3ab.log
tools/r04_k3_ab.sh 20 20
python tools/gpu_parity_at_scale.py 2e6 >> gpurun_out/r04_k1win.log 2>&1

timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_two_ranks.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > gpurun_out/r04_prunepipe.log
rm -f gpurun_out/k3ab.log
echo "== PRUNE pipelined" >> gpurun_out/k3ab.log; tools/r04_k3_ab.sh 20
echo "== W2RAP_PRUNE_ONE=1" >> gpurun_out/k3ab.log; W2RAP_PRUNE_ONE=1 tools/r04_k3_ab.sh 20
echo "== PRUNE pipelined" >> gpurun_out/k3ab.log; tools/r04_k3_ab.sh 20

timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or hash_classes" 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -12 > gpurun_out/r04_fused.log
W2RAP_FUSED_PRUNE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 >> gpurun_out/r04_fused.log
rm -f gpurun_out/k3ab.log
echo "== FUSED=1" >> gpurun_out/k3ab.log; W2RAP_FUSED_PRUNE=1 tools/r04_k3_ab.sh 20
echo "== FUSED=0" >> gpurun_out/k3ab.log; tools/r04_k3_ab.sh 20
echo "== FUSED=1" >> gpurun_out/k3ab.log; W2RAP_FUSED_PRUNE=1 tools/r04_k3_ab.sh 20

timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > gpurun_out/r04_parity3.log
rm -f gpurun_out/k3ab.log
tools/r04_k3_ab.sh 22 24 20
echo "== KPB=4500" >> gpurun_out/k3ab.log; W2RAP_KPB=4500 tools/r04_k3_ab.sh 22
echo "== KPB=5500" >> gpurun_out/k3ab.log; W2RAP_KPB=5500 tools/r04_k3_ab.sh 24

timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "shapes or stages or overflow or heavy or replay" 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -5 > gpurun_out/r04_parity2.log
rm -f gpurun_out/k3ab.log
tools/r04_k3_ab.sh 22 20
echo "== KPB=4500" >> gpurun_out/k3ab.log; W2RAP_KPB=4500 tools/r04_k3_ab.sh 22
echo "== KPB=6000" >> gpurun_out/k3ab.log; W2RAP_KPB=6000 tools/r04_k3_ab.sh 20

timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > gpurun_out/r04_pytest_gpu.log
rm -f gpurun_out/k3ab.log
tools/r04_k3_ab.sh 20
for c in 8 32; do echo "== K1_CHUNK=$c" >> gpurun_out/k3ab.log; W2RAP_K1_CHUNK=$c tools/r04_k3_ab.sh 20; done
for c in 2 8; do echo "== SLICES=$c" >> gpurun_out/k3ab.log; W2RAP_SLICES=$c tools/r04_k3_ab.sh 20; done

timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -x -q -k "internal_retry or replay or stages or long or 251 or quirk or tail or second_build or genomes or properties" 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > gpurun_out/r04_parity5.log
python tools/gpu_planted_check.py 1e6 2e6 5e7 > gpurun_out/r04_planted_wave.txt 2>&1
W2RAP_PATH_BUDGET=6 python tools/gpu_planted_check.py 5e7 > gpurun_out/r04_planted_wave_b6.txt 2>&1
W2RAP_PATH_BUDGET=4 python tools/gpu_planted_check.py 5e7 > gpurun_out/r04_planted_wave_b4.txt 2>&1
W2RAP_PATH_WAVE=0 python tools/gpu_planted_check.py 5e7 > gpurun_out/r04_planted_lane.txt 2>&1
rm -f gpurun_out/k3ab.log
echo "== KPB=4500" >> gpurun_out/k3ab.log; W2RAP_KPB=4500 tools/r04_k3_ab.sh 20 24
echo "== KPB=4000" >> gpurun_out/k3ab.log; W2RAP_KPB=4000 tools/r04_k3_ab.sh 24
echo "== BATCHES=6" >> gpurun_out/k3ab.log; W2RAP_BATCHES=6 tools/r04_k3_ab.sh 24
echo "== BATCHES=8" >> gpurun_out/k3ab.log; W2RAP_BATCHES=8 tools/r04_k3_ab.sh 24
W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --reads 62.5e6 --genome 312.5e6 --no-cpu-baseline --no-extras > gpurun_out/r04_dist_world1.json 2> gpurun_out/r04_dist_world1.err

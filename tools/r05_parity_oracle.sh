#!/bin/bash
# parity of the final library against the single-threaded oracle at sizes beyond the test suite's: four coverage regimes at 2-6 M reads, then
# BASELINE configs[1] itself (50 M reads; the oracle alone takes ~25 minutes)
mkdir -p gpurun_out
( python3 tools/gpu_parity_at_scale.py 2e6 4e5; python3 tools/gpu_parity_at_scale.py 4e6 4e7; python3 tools/gpu_parity_at_scale.py 6e6 3e7; python3 tools/gpu_parity_at_scale.py 2e6 2e6 ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp" > gpurun_out/r05_parity_scale.txt
( time python3 tools/gpu_parity_at_scale.py 5e7 2.5e8 step3 ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp" > gpurun_out/r05_parity_50M_reads.txt
cat gpurun_out/r05_parity_scale.txt gpurun_out/r05_parity_50M_reads.txt

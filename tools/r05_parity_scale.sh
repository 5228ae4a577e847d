#!/bin/bash
# the sharded path at sizes beyond the test suite's: 4 M reads against the oracle (3 ranks, hooks), 50 M reads (BASELINE configs[1]) with 2 and 4
# ranks sharing the GPU against the one-GPU path, plain and with the hooks
mkdir -p gpurun_out
( python3 tools/gpu_sharded_at_scale.py 4e6 2e7 3 5 8 5e6
  python3 tools/gpu_sharded_at_scale.py 5e7 2.5e8 2
  python3 tools/gpu_sharded_at_scale.py 5e7 2.5e8 4 27 8 ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|WARNING: test hook" > gpurun_out/r05_parity_sharded.txt
cat gpurun_out/r05_parity_sharded.txt

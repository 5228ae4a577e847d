#!/bin/bash
# the multi-GPU default path at sizes beyond the test suite's, on HEAD: against the oracle (4 M uniform reads, 3 ranks, hooks), against the
# one-GPU path (50 M uniform reads, 2 and 4 ranks), and -- planted generator -- against the REAL reference at 16 M and 50 M reads
mkdir -p gpurun_out
export TMPDIR=/tmp
( python3 tools/gpu_sharded_at_scale.py 4e6 2e7 3 5 8 5e6
  python3 tools/gpu_sharded_at_scale.py 5e7 2.5e8 2
  python3 tools/gpu_sharded_at_scale.py 5e7 2.5e8 4 27 8
  timeout 1200 python3 tools/gpu_parity_ref_scale.py 16e6 1 4
  timeout 2400 python3 tools/gpu_parity_ref_scale.py 50e6 1 4 ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|WARNING: test hook\|^{" > gpurun_out/r06_parity_sharded.txt
cat gpurun_out/r06_parity_sharded.txt
# Step 3 on HEAD: the entry behind Step 2 (W2RAP_STEP3_UNIQUE_KMERS: K2-mers strictly inside an unshared edge skip the dictionary) against the
# host-buffer entry WITHOUT the flag (every K2-mer grouped by content) at 50 M diploid reads, and both against the oracle at 4 M
( python3 tools/gpu_step3_scale.py 4e6 2000 1
  python3 tools/gpu_step3_scale.py 4e6 300 1
  python3 tools/gpu_step3_scale.py 5e7 2000 ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|WARNING: test hook\|^{" > gpurun_out/r06_parity_step3.txt
cat gpurun_out/r06_parity_step3.txt

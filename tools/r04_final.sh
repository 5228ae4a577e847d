#!/bin/bash
# The round's final set on one box: whole GPU test suite, the default bench line (with cpu_baseline parity verdict and extras), the same
# command under rocprofv3 --kernel-trace --stats, and the PMC passes.  Outputs under gpurun_out/final4/.
out=gpurun_out/final4; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/prof -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_under_rocprofv3.json 2> $out/prof.err
python3 tools/kernel_stats_md.py $out/prof $out/kernel_stats.md $out/kernel_stats_w2.csv
rm -rf $out/prof
tools/r04_pmc.sh final
cp gpurun_out/pmc_final.md $out/pmc.md; cp gpurun_out/pmc_final.json $out/pmc_step2.json
rm -rf gpurun_out/pmc_final
# the distributed code path at world 1 on the per-GPU share of configs[2] (the scale model's inputs)
W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 timeout 900 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/dist_world1.json 2> $out/dist_world1_trace.txt

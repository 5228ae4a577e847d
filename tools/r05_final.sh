#!/bin/bash
# The round's final set on one box: whole GPU test suite, the default bench line (with cpu_baseline parity verdicts and extras), the same
# command under rocprofv3 --kernel-trace --stats, the PMC passes, and the sharded code path forced to run at world 1 -- plain and with the
# two test hooks that give one GPU the list sizes of an 8-rank job (the scale model's inputs).  Outputs under gpurun_out/final5/.
out=gpurun_out/final5; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8 > $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/prof -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_under_rocprofv3.json 2> $out/prof.err
python3 tools/kernel_stats_md.py $out/prof $out/kernel_stats.md $out/kernel_stats_w2.csv
rm -rf $out/prof
tools/r04_pmc.sh final
cp gpurun_out/pmc_final.md $out/pmc.md; cp gpurun_out/pmc_final.json $out/pmc_step2.json
rm -rf gpurun_out/pmc_final
for cfg in "0 0" "27 8" "3 0"; do
  set -- $cfg
  unset W2RAP_TEST_SHARD_CUT W2RAP_TEST_SHARD_VIRTUAL
  if [ $1 != 0 ]; then export W2RAP_TEST_SHARD_CUT=$1; fi
  if [ $2 != 0 ]; then export W2RAP_TEST_SHARD_VIRTUAL=$2; fi
  name=dist_world1; if [ "$cfg" != "0 0" ]; then name=dist_world1_cut$1_v$2; fi
  W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 timeout 900 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.json 2> $out/${name}_trace.txt
done

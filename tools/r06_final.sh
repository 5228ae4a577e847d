#!/bin/bash
# The round's final set on one box: whole GPU test suite, smoke, the default bench line (cpu_baseline parity verdicts incl. the index and the
# sharded route, extras), the same command under rocprofv3 --kernel-trace --stats, the PMC passes, the sharded code path forced at world 1 --
# plain and with the hooks that give one GPU the list sizes of an 8-rank job -- and the one-GPU path on the same 62.5 M reads.
out=gpurun_out/final6; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -22 > $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/prof -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_under_rocprofv3.json 2> $out/prof.err
python3 tools/kernel_stats_md.py $out/prof $out/kernel_stats.md $out/kernel_stats_w2.csv
rm -rf $out/prof
tools/r04_pmc.sh final6
cp gpurun_out/pmc_final6.md $out/pmc.md; cp gpurun_out/pmc_final6.json $out/pmc_step2.json
rm -rf gpurun_out/pmc_final6
for cfg in "0 0" "27 8"; do
  set -- $cfg
  unset W2RAP_TEST_SHARD_CUT W2RAP_TEST_SHARD_VIRTUAL
  if [ $1 != 0 ]; then export W2RAP_TEST_SHARD_CUT=$1; fi
  if [ $2 != 0 ]; then export W2RAP_TEST_SHARD_VIRTUAL=$2; fi
  name=dist_world1; if [ "$cfg" != "0 0" ]; then name=dist_world1_cut$1_v$2; fi
  W2RAP_FORCE_DIST=1 timeout 900 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.json 2> $out/${name}.err
done
unset W2RAP_TEST_SHARD_CUT W2RAP_TEST_SHARD_VIRTUAL
W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 W2RAP_TRACE_SHARD=1 timeout 900 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/dist_world1_trace.txt
timeout 600 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/one_gpu_62M.json 2> $out/one_gpu_62M.err
# Step 3 behind Step 2 (lone places, block index: round 6), the same under rocprofv3 with its timeline, and Steps 1 -> 2 -> 3 chained
timeout 600 python bench.py --step3 --no-cpu-baseline > $out/bench_step3.json 2> $out/bench_step3.err
W2RAP_STEP3_NO_LONE=1 timeout 600 python bench.py --step3 --no-cpu-baseline > $out/bench_step3_no_lone.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $out/prof3 -o x --output-format csv -- python3 bench.py --step3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_step3_under_rocprofv3.json 2> $out/prof3.err
python3 tools/kernel_stats_md.py $out/prof3 $out/step3_kernel_stats.md $out/step3_kernel_stats.csv
python3 tools/kernel_timeline.py $out/prof3 k3_obj_len k3_read_paths 12 > $out/step3_timeline.txt 2>&1
rm -rf $out/prof3
timeout 900 python bench.py --pipeline --no-cpu-baseline > $out/bench_pipeline.json 2> $out/bench_pipeline.err
tail -3 $out/pytest_gpu.log; cat $out/smoke.log | tail -3
python3 - <<'PY'
import json
for n in ("bench","bench_under_rocprofv3","dist_world1","dist_world1_cut27_v8","one_gpu_62M","bench_step3","bench_step3_no_lone","bench_pipeline"):
    try:
        d=json.loads(open(f"gpurun_out/final6/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],1), {k:round(v,1) for k,v in d.get("phase_ms", d.get("stage_ms", {})).items()}, "frac", round(d.get("roofline",{}).get("frac",0),3))
    except Exception as e: print(n, "failed", e)
PY

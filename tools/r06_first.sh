#!/bin/bash
# round 6, first measurements on HEAD: the default bench line (new keys: planted verdicts through the index and the sharded path, binned PQVec,
# copy forms), the sharded code path forced at world 1 (plain and with the hooks), the 8-waves-per-SIMD shape of the counting kernel
out=gpurun_out/r06a; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err
for cfg in "0 0" "27 8"; do
  set -- $cfg
  unset W2RAP_TEST_SHARD_CUT W2RAP_TEST_SHARD_VIRTUAL
  if [ $1 != 0 ]; then export W2RAP_TEST_SHARD_CUT=$1; fi
  if [ $2 != 0 ]; then export W2RAP_TEST_SHARD_VIRTUAL=$2; fi
  name=dist_world1; if [ "$cfg" != "0 0" ]; then name=dist_world1_cut$1_v$2; fi
  W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 W2RAP_TRACE_SHARD=1 timeout 900 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.json 2> $out/${name}_trace.txt
done
unset W2RAP_TEST_SHARD_CUT W2RAP_TEST_SHARD_VIRTUAL
timeout 600 python bench.py --reads 62.5e6 --genome 312.5e6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/one_gpu_62M.json 2> $out/one_gpu_62M.err
for k3 in 22 21 22 21; do
  W2RAP_K3=$k3 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('W2RAP_K3=$k3: step %.1f ms, count %.1f, k_count_fp %.2f, k_table_insert %.2f' % (d['ms_per_step'], d['phase_ms']['count'], k.get('k_count_fp',0), k.get('k_table_insert',0)))" >> $out/k3_occupancy_ab.txt
done
cat $out/k3_occupancy_ab.txt
python3 - <<'PY'
import json
for n in ("bench","dist_world1","dist_world1_cut27_v8","one_gpu_62M"):
    try:
        d=json.loads(open(f"gpurun_out/r06a/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["phase_ms"].items()}, d.get("exchange_ms"))
    except Exception as e: print(n, "failed", e)
PY

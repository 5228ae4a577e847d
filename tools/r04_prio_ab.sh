#!/bin/bash
# stream priorities x LDS tile stride of k_count_fp: default build and a -DW2RAP_FP_TS9 build (odd stride), main stream above / below / level with the side stream
mkdir -p gpurun_out; rm -f gpurun_out/k3ab.log
mkdir -p /tmp/ts9; cd w2rap_contigger_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DW2RAP_FP_TS9 -c step2_count.hip -o /tmp/ts9/step2_count.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ts9/libw2rap_step2.so /tmp/ts9/step2_count.o build/step2_graph.o build/step2_path.o build/step2_prims.o build/step2_capi.o build/step2_run.o build/step3_repath.o build/step1_ingest.o build/gfa_dump.o -lpthread 2>/dev/null; cd ../..
for lib in "" /tmp/ts9/libw2rap_step2.so; do
  if [ -n "$lib" ]; then export W2RAP_LIB=$lib; else unset W2RAP_LIB; fi
  for cfg in 20 22; do
    echo "== lib=${lib:-default} cfg=$cfg main above side" >> gpurun_out/k3ab.log; bash tools/r04_k3_ab.sh $cfg
    echo "== lib=${lib:-default} cfg=$cfg FLIP" >> gpurun_out/k3ab.log; W2RAP_PRIO_FLIP=1 bash tools/r04_k3_ab.sh $cfg
    echo "== lib=${lib:-default} cfg=$cfg SAME" >> gpurun_out/k3ab.log; W2RAP_PRIO_SAME=1 bash tools/r04_k3_ab.sh $cfg
  done
done

#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/k1_diag.txt
for b in 1 ""; do
for lib in "" 1 2 3 4; do
  if [ -n "$lib" ]; then export W2RAP_LIB=$PWD/tools/diag/libdiag$lib.so; else unset W2RAP_LIB; fi
  W2RAP_BATCHES=$b timeout 300 python tools/diag/k1_diag.py >> gpurun_out/k1_diag.txt 2>gpurun_out/k1_diag_$lib.err
done; done

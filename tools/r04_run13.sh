rm -f gpurun_out/k3ab.log
for f in 0.6 1.0 0.4 0.6 1.0; do echo "== LAST_BATCH=$f" >> gpurun_out/k3ab.log; W2RAP_LAST_BATCH=$f tools/r04_k3_ab.sh 20; done

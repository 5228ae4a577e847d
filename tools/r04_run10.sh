rm -f gpurun_out/k3ab.log
for c in 8 6 7 8 6; do echo "== K1_MINW=$c" >> gpurun_out/k3ab.log; W2RAP_K1_MINW=$c tools/r04_k3_ab.sh 20; done

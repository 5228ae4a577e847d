NO_TRAFFIC=1 tools/r04_pmc.sh k3_22 W2RAP_K3=22
for kpb in 4000 4500; do echo "== KPB=$kpb cfg 22" >> gpurun_out/k3ab.log; W2RAP_KPB=$kpb tools/r04_k3_ab.sh 22; done
echo "== KPB=4500 cfg 20" >> gpurun_out/k3ab.log; W2RAP_KPB=4500 tools/r04_k3_ab.sh 20

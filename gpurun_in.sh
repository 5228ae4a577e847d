timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "0 5000" "4 5000" "4 6500" "5 5000" "2 5000"; do set -- $cfg; echo "== K3=$1 KPB=$2"; W2RAP_NO_OVERLAP=1 W2RAP_K3=$1 W2RAP_KPB=$2 W2RAP_TRACE=1 timeout 60 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "count:" | tail -1 | cut -c1-500; done
W2RAP_K3=4 timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

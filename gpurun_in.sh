timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for cfg in "9 5000" "0 5000" "0 6000" "0 4000"; do set -- $cfg; echo "== K3=$1 KPB=$2"; W2RAP_K3=$1 W2RAP_KPB=$2 W2RAP_TRACE=1 timeout 60 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "count:|clocks" | tail -3 | cut -c1-500; done

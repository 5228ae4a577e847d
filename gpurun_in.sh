W2RAP_PATH_PROF=1 W2RAP_TRACE=1 timeout 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "k_path clocks|metric" | tail -2 | grep -o 'k_path clocks.*\|"kernel_ms_per_step.*'

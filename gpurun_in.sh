timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "metric" | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"phase_ms.*"kmers_per_s_count\|"kernel_ms_per_step.*'

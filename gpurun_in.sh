timeout 600 python -m pytest tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4

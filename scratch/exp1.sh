python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py -m gpu -q 2>&1 | grep -E "passed|failed"
GDF_JK_WIDE=1 python -m pytest tests/test_gpu_join.py -m gpu -q 2>&1 | grep -E "passed|failed"

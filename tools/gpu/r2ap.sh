ulimit -c 0
GDF_STRESS_SECONDS=150 timeout 1200 python -m pytest tests/test_gpu_stress.py -x -q -m gpu -p no:cacheprovider -k group_by 2>&1 | grep -E "^E|tag|assert" | head -30

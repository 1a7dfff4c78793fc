O=gpurun_out/r2aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_join.py tests/test_gpu_join.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt

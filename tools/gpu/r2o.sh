O=gpurun_out/r2o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py tests/test_gpu_sort.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python tools/bench_c5.py 2>>$O/err.txt | cut -c1-700

O=gpurun_out/r2m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_groupby.py tests/test_gpu_c5.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | cut -c1-100,600-1500
timeout 600 python tools/bench_c5.py 2>>$O/err.txt | cut -c1-900
GDF_GB_NO_FUSED=1 timeout 600 python tools/bench_c5.py 2>>$O/err.txt | cut -c1-600
bash tools/gpu/gaps.sh 2>&1 | grep "make_units\|total gap"

O=gpurun_out/r2n; mkdir -p $O
timeout 600 python tools/bench_c5.py 2>>$O/err.txt | cut -c1-700
timeout 600 python tools/bench_ops.py --ops groupby --groups 1000000 2>>$O/err.txt | cut -c1-500
timeout 600 python tools/bench_ops.py --ops groupby --groups 10000000 2>>$O/err.txt | cut -c1-500
GDF_GB_NO_FUSED=1 timeout 600 python tools/bench_ops.py --ops groupby --groups 1000000 2>>$O/err.txt | cut -c1-500
GDF_GB_NO_FUSED=1 timeout 600 python tools/bench_ops.py --ops groupby --groups 10000000 2>>$O/err.txt | cut -c1-500
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -x -q -m gpu -k "not full_size and not 2_to_the_29" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt

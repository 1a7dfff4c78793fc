O=gpurun_out/r2j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_multirank_one_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | cut -c1-100,600-1500
GDF_JK_NO_DEFER=1 timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | cut -c1-100,600-1500
timeout 900 python tools/bench_shapes.py --only c3_zipf_probe,c3_half_hit,c3_left_half_hit,c3_int32_keys > $O/shapes.jsonl 2>>$O/err.txt; cut -c1-60,280-900 $O/shapes.jsonl
bash tools/gpu/gaps.sh 2>&1 | tail -32

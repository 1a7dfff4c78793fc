O=gpurun_out/r2p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_multirank_one_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python tools/bench_shapes.py --only dup4_build_keys,c3_half_hit,c3_headline > $O/shapes.jsonl 2>$O/err.txt; cut -c1-60,280-900 $O/shapes.jsonl
timeout 600 python tools/bench_c5.py 2>>$O/err.txt | cut -c1-500

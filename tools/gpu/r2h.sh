O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_groupby.py tests/test_gpu_hash_partition.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python tools/bench_shapes.py --only c3_wide_keys,c2_sparse_keys,c2_dense_keys,c3_headline > $O/shapes.jsonl 2>$O/err.txt; cut -c1-120,280-1000 $O/shapes.jsonl
for b in 15 17 18; do echo dict_bits=$b; GDF_GB_DICT_BITS=$b timeout 300 python tools/bench_shapes.py --only c2_sparse_keys 2>>$O/err.txt | cut -c300-900; done

set -x
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hash_partition.py tests/test_gpu_filter.py -x -q -m gpu -k "prefixsum or stencil or filter" > $O/pytest_scan.txt 2>&1; tail -5 $O/pytest_scan.txt
timeout 300 python tools/bench_ops.py --ops scan > $O/scan_lb.jsonl 2>$O/err.txt; cat $O/scan_lb.jsonl
GDF_SCAN_3PASS=1 timeout 300 python tools/bench_ops.py --ops scan > $O/scan_3p.jsonl 2>>$O/err.txt; cat $O/scan_3p.jsonl
timeout 600 python tools/bench_shapes.py --only c3_zipf_probe,dup4_build_keys,c3_headline > $O/shapes.jsonl 2>>$O/err.txt; cut -c1-600 $O/shapes.jsonl
timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py -x -q -m gpu > $O/pytest_join.txt 2>&1; tail -5 $O/pytest_join.txt

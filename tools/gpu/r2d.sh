set -x
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hash_partition.py -x -q -m gpu -k "prefixsum" > $O/pytest_scan.txt 2>&1; tail -3 $O/pytest_scan.txt
timeout 300 python tools/bench_ops.py --ops scan > $O/scan_lb.jsonl 2>$O/err.txt; cat $O/scan_lb.jsonl
timeout 900 python -m pytest tests/test_gpu_join.py -x -q -m gpu -k "materialisation" > $O/pytest_mat.txt 2>&1; tail -5 $O/pytest_mat.txt
timeout 600 python tools/bench_shapes.py --only c3_materialise_2_payload_cols > $O/shapes.jsonl 2>>$O/err.txt; cut -c1-900 $O/shapes.jsonl

O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hash_partition.py -x -q -m gpu -k "prefixsum" > $O/pytest_scan.txt 2>&1; tail -3 $O/pytest_scan.txt
for d in 0 2; do for w in 0 1 3; do echo dbg=$d wgs=$w; GDF_SCAN_WGS_PER_CU=$w GDF_SCAN_DBG=$d timeout 300 python tools/bench_ops.py --ops scan 2>>$O/err.txt | cut -c100-400; done; done
timeout 900 python -m pytest tests/test_gpu_join.py -x -q -m gpu > $O/pytest_join.txt 2>&1; tail -3 $O/pytest_join.txt
timeout 600 python tools/bench_shapes.py --only c3_wide_keys > $O/shapes.jsonl 2>>$O/err.txt; cut -c1-900 $O/shapes.jsonl

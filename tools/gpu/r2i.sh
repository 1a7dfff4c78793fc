O=gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hash_partition.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/bench_ops.py --ops partition,scan 2>$O/err.txt | cut -c1-330
GDF_JK_DBG=512 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pandas-sample 0 2>&1 | grep -v "^{" | tail -40
bash tools/gpu/gaps.sh 2>&1 | tail -25

O=gpurun_out/r2ae; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_hash_partition.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/bench_shapes.py --only c2_dense_keys,c2_sparse_keys --reps 10 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"

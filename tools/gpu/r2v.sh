O=gpurun_out/r2v; mkdir -p $O
timeout 600 python tools/bench_shapes.py --only c2_dense_keys,c2_sparse_keys 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python tools/bench_ops.py 2>>$O/err.txt | tail -12

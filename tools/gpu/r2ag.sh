O=gpurun_out/r2ag; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_hash_partition.py tests/test_gpu_c5.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python tools/bench_shapes.py --only c3_headline,c3_half_hit 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
GDF_JK_NO_SPARSE_OPT=1 timeout 600 python tools/bench_shapes.py --only c3_half_hit 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('NO_SPARSE', d['shape'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"

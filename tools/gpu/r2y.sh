O=gpurun_out/r2y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_join.py -x -q -m gpu -k "material or result or payload" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python tools/bench_shapes.py --only c3_materialise_2_payload_cols 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
tail -3 $O/err.txt

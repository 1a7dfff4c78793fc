O=gpurun_out/r2aj; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3))"
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench default-like', round(d['ms_per_step'],3))"
for i in 1 2; do
timeout 900 python tools/bench_shapes.py --only c3_headline,c3_half_hit,c3_zipf_probe,dup4_build_keys --reps 5 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(sum(d['kernels_ms'].values()),3))"
done
timeout 900 python -m pytest tests/test_rmm.py tests/test_gpu_join.py -x -q -m gpu 2>&1 | tail -2

O=gpurun_out/r2ak; mkdir -p $O
timeout 900 python tools/bench_shapes.py --only c3_half_hit,c3_tenth_hit --reps 5 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(sum(d['kernels_ms'].values()),3), {k: v for k,v in d['kernels_ms'].items() if v > 0.2})"
GDF_JK_NO_SPARSE_OPT=1 timeout 900 python tools/bench_shapes.py --only c3_tenth_hit --reps 5 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('NO_SPARSE', d['shape'], round(d['ms'],3), round(sum(d['kernels_ms'].values()),3), {k: v for k,v in d['kernels_ms'].items() if v > 0.2})"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3

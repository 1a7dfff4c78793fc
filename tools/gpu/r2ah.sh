O=gpurun_out/r2ah; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v > 0.2})"
for i in 1 2; do
timeout 600 python tools/bench_shapes.py --only c3_headline,c3_half_hit --reps 5 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), {k: v for k,v in d['kernels_ms'].items() if v > 0.2})"
done
GDF_JK_NO_SPARSE_OPT=1 timeout 600 python tools/bench_shapes.py --only c3_headline,c3_half_hit --reps 5 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('NO_SPARSE', d['shape'], round(d['ms'],3), {k: v for k,v in d['kernels_ms'].items() if v > 0.2})"

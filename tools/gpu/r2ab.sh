O=gpurun_out/r2ab; mkdir -p $O
for d in 0 8 16 24; do
  echo "SDBG=$d"; GDF_JK_SDBG=$d timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v > 0.2})"
done

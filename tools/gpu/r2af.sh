O=gpurun_out/r2af; mkdir -p $O
for b in 256 128 64 16; do
  echo "BIG_FROM=$b"; GDF_HP_BIG_FROM=$b timeout 300 python tools/bench_ops.py --ops partition 2>>$O/err.txt | grep hash_partition | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['op'][-8:], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
done

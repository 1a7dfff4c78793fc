python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py -m gpu -q 2>&1 | grep -E "passed|failed"
GDF_JK_WIDE=1 python -m pytest tests/test_gpu_join.py -m gpu -q 2>&1 | grep -E "passed|failed"
python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['ms_per_step'], d['value']/1e9, {x:round(k[x],2) for x in k if k[x]>0.05})"

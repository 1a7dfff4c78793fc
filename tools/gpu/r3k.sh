# round 3, call k: seven-byte level-1 tuples: parity + A/B (LAB build: GDF_JK_NO_P7 / GDF_JK_NO_P6 are path switches read from the environment there)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py -m gpu -x -q -k "six_byte or headline or speculative or masked_single or random_values or skew" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_join.txt
cat $O/pytest_join.txt
for i in 1 2 3; do
  LIBGDF_AMD_LAB=1 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p7+p6', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/ab.txt
  LIBGDF_AMD_LAB=1 GDF_JK_NO_P7=1 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p6   ', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/ab.txt
done
cat $O/ab.txt

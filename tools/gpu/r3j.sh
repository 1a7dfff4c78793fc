# round 3, call j: pairwise 12-byte stores of the six-byte tuples, the lean multimap kernel: parity + benches
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_stress.py -m gpu -x -q -k "not 2_to_the_29" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_join.txt
cat $O/pytest_join.txt
for i in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p6  ', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/ab.txt
  LIBGDF_AMD_LAB=1 GDF_JK_NO_P6=1 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('8byte', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/ab.txt
done
cat $O/ab.txt
python tools/bench_shapes.py --only dup4_build_keys > $O/bench_shapes.jsonl 2>$O/bench_shapes.err
LIBGDF_AMD_LAB=1 GDF_JK_NO_MULTI=1 python tools/bench_shapes.py --only dup4_build_keys >> $O/bench_shapes.jsonl 2>>$O/bench_shapes.err
python - <<'PY'
import json
for l in open('gpurun_out/r3j/bench_shapes.jsonl'):
    d=json.loads(l); print(d['shape'], round(d['ms'],2), d['out_rows'], d['kernels_ms'])
PY

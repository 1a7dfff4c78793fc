# C3: jk_probe_fast with one output claim per wave and batch (default) against one per tuple (GDF_JK_DBG=2048), alternating; join parity
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bl; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_fused_join.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -10 > $O/pytest_join.txt
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 > $O/bench_new_$i.json
  GDF_JK_DBG=2048 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 > $O/bench_old_$i.json
done
python tools/bench_shapes.py --only c3_half_hit,c3_left_half_hit,c3_wide_keys,dup4_build_keys > $O/shapes.jsonl 2>/dev/null
timeout 300 python tools/stress_join.py --seconds 60 --seed 201 > $O/stress201.txt 2>&1
cat $O/pytest_join.txt; for f in new_1 old_1 new_2 old_2; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})"; done
python -c "
import json
for l in open('$O/shapes.jsonl'):
    x=json.loads(l); print(x['shape'], round(x['ms'],2), x['kernels_ms'])
"; tail -n 1 $O/stress201.txt

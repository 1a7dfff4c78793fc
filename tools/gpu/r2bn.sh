# general probe kernel (jk_probe<WRITE>): one claim per wave and batch (default) against one per tuple slot (GDF_JK_DBG=2048); join parity + stress
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bn; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_fused_join.py tests/test_gpu_multirank_one_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -10 > $O/pytest_join.txt
python tools/bench_shapes.py --only dup4_build_keys,c3_left_half_hit,c3_wide_keys,c3_half_hit > $O/shapes_new.jsonl 2>/dev/null
GDF_JK_DBG=2048 python tools/bench_shapes.py --only dup4_build_keys,c3_left_half_hit,c3_wide_keys,c3_half_hit > $O/shapes_old.jsonl 2>/dev/null
timeout 300 python tools/stress_join.py --seconds 100 --seed 404 > $O/stress404.txt 2>&1
cat $O/pytest_join.txt
python -c "
import json
for f in ('new','old'):
    for l in open('$O/shapes_%s.jsonl' % f):
        x=json.loads(l); print(f, x['shape'], round(x['ms'],2), {k:v for k,v in x['kernels_ms'].items() if v>0.2})
"; tail -n 1 $O/stress404.txt

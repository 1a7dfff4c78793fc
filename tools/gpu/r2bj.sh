# every small device -> host read-back of groupby / filter / sort through the pinned staging buffer (tools/readback_lab.hip: 35 -> 15 us each)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bj; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | head -5 > $O/pytest_gpu.txt
python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5.json
python tools/bench_ops.py > $O/bench_ops.jsonl 2>/dev/null
python tools/bench_shapes.py --only c2_dense_keys > $O/shapes_c2.jsonl 2>/dev/null
python tools/bench_shapes.py --only c2_sparse_keys >> $O/shapes_c2.jsonl 2>/dev/null
cat $O/pytest_gpu.txt; python -c "import json,sys; d=json.load(open('$O/c5.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; python - <<'P'
import json
for l in open('gpurun_out/r2bj/bench_ops.jsonl'):
    x=json.loads(l); print(x['op'][:62], round(x['ms'],3))
for l in open('gpurun_out/r2bj/shapes_c2.jsonl'):
    x=json.loads(l); print(x['shape'], round(x['ms'],3))
P

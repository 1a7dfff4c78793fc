# small host-side savings on the direct path (C2) + 12 rows per thread in the single-column count pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bi; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5.json
python tools/bench_ops.py --ops groupby > $O/bench_ops_groupby.jsonl 2>/dev/null
python tools/bench_shapes.py --only c2_dense_keys > $O/shapes_c2.jsonl 2>/dev/null
python tools/bench_shapes.py --only c2_sparse_keys >> $O/shapes_c2.jsonl 2>/dev/null
cat $O/pytest_groupby.txt; python -c "import json,sys; d=json.load(open('$O/c5.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; cut -c1-330 $O/bench_ops_groupby.jsonl; cut -c1-300 $O/shapes_c2.jsonl

# C5: the count pass leaves out the low key column (GDF_GBP_COUNT_ALL=1 reads it as before); parity of the whole suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bf; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
for i in 1 2; do
  python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_skip_$i.json
  GDF_GBP_COUNT_ALL=1 python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_all_$i.json
done
python tools/bench_ops.py --ops groupby > $O/bench_ops_groupby.jsonl 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | head -5 > $O/pytest_gpu.txt
cat $O/pytest_groupby.txt $O/pytest_gpu.txt; for f in skip_1 all_1 skip_2 all_2; do echo $f; python -c "import json,sys; d=json.load(open('$O/c5_$f.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; done

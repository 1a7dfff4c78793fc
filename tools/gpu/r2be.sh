# C5: batched flush reads (now default) and the next-tile line touches (GDF_GBP_PREFETCH=1), same box, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2be; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
for i in 1 2; do
  python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_default_$i.json
  GDF_GBP_PREFETCH=1 python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_prefetch_$i.json
done
cat $O/pytest_groupby.txt; for f in default_1 prefetch_1 default_2 prefetch_2; do echo $f; python -c "import json,sys; d=json.load(open('$O/c5_$f.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; done

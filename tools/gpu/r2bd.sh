# C5: hybrid ranking (two ballot-settled leaders + plain LDS atomics) against the 11-bit match-any, same box, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bd; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py "tests/test_gpu_stress.py::test_group_by_random_shapes_against_the_oracle" -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
for i in 1 2; do
  python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_hybrid_$i.json
  GDF_GBP_RANK_MATCH=1 python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_match_$i.json
done
cat $O/pytest_groupby.txt; for f in hybrid_1 match_1 hybrid_2 match_2; do echo $f; python -c "import json,sys; d=json.load(open('$O/c5_$f.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; done

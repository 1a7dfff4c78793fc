# parity after the generic gbp_scatter took gbp_rank: group-by suite, a longer random-shape stress, C5 through the generic kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bh; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
GDF_STRESS_SECONDS=150 timeout 900 python -m pytest "tests/test_gpu_stress.py::test_group_by_random_shapes_against_the_oracle" -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_stress.txt
GDF_GBP_OLD=1 python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_generic.json
GDF_GBP_DYNAMIC=1 python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_dynamic.json
python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_default.json
cat $O/pytest_groupby.txt $O/pytest_stress.txt; for f in generic dynamic default; do echo $f; python -c "import json,sys; d=json.load(open('$O/c5_$f.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; done
